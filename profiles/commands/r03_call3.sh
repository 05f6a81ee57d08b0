#!/bin/bash
# round-3 GPU call 3: RWKV7 wide step test, bench --backend rwkv7, RWKV lane shapes, B = 1 decode step kernel trace (durations and gaps),
# gemm_mid tile order A/B, rocprofv3 --pmc without --step-sync now that the library bounds its queue depth
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3c3; mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_round3.py tests/test_gpu_model.py tests/test_gpu_round2.py -m gpu -x -q -k "rwkv or wide_group" > $O/pytest_rwkv.log 2>&1; rc=$?
tail -5 $O/pytest_rwkv.log; echo "pytest rwkv rc=$rc"
[ $rc -ne 0 ] && exit 1
timeout -k 10 600 python bench.py --backend rwkv7 --steps 2 --warmup 1 > $O/bench_rwkv7.json 2> $O/bench_rwkv7.err; rc=$?
tail -3 $O/bench_rwkv7.err; cut -c1-900 $O/bench_rwkv7.json; echo "bench rwkv7 rc=$rc"
[ $rc -ne 0 ] && exit 1
for shape in "384 2048 128 3" "512 2048 256 2" "512 2048 512 1" "1024 2048 512 2" "1024 2048 1024 1"; do
  timeout -k 10 300 python tools/bench_rwkv7.py $shape >> $O/rwkv7_shapes.jsonl 2>> $O/rwkv7_shapes.err || { echo "shape $shape failed"; tail -3 $O/rwkv7_shapes.err; }
done
cat $O/rwkv7_shapes.jsonl | cut -c1-400
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/b1_trace -- python3 $R/tools/bench_decode_step.py --iters 20 --max-streams 1 "b1:B=1,L=1,g=1" > $O/b1_trace.out 2> $O/b1_trace.err; rc=$?
echo "b1 trace rc=$rc"; tail -2 $O/b1_trace.out | cut -c1-300
python3 $R/tools/trace_gaps.py $O/b1_trace --last 1550 > $O/b1_gaps.txt 2>&1; head -30 $O/b1_gaps.txt; rm -rf $O/b1_trace
cd $R
timeout -k 10 400 python bench.py --steps 4 --warmup 1 --cpu-sample 0 --profile-tokens 0 --overlap 0 > $O/bench_colmaj_auto.json 2> $O/bench_colmaj_auto.err; rc=$?
python -c "import json;d=json.loads(open('$O/bench_colmaj_auto.json').read().strip().splitlines()[-1]);print('colmaj auto', d['value'], d['compress_tokens_per_s_rank0'], d['decompress_tokens_per_s_rank0'])"
timeout -k 10 400 python bench.py --steps 4 --warmup 1 --cpu-sample 0 --profile-tokens 0 --overlap 0 --debug-set mid_colmaj=0 > $O/bench_colmaj_0.json 2> $O/bench_colmaj_0.err; rc=$?
python -c "import json;d=json.loads(open('$O/bench_colmaj_0.json').read().strip().splitlines()[-1]);print('colmaj 0   ', d['value'], d['compress_tokens_per_s_rank0'], d['decompress_tokens_per_s_rank0'])"
cd /tmp
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_nosync -- python3 $R/bench.py --streams 128 --lanes 1 --lane-streams 128 --no-graph --no-events --overlap 0 --sample-tokens 1536 --window 1536 --steps 1 --warmup 0 --cpu-sample 0 --profile-tokens 0 > $O/pmc_nosync.out 2> $O/pmc_nosync.err; rc=$?
echo "pmc (no --step-sync, library bounds its queue depth) rc=$rc"; grep -v "^    @" $O/pmc_nosync.err | tail -4 | cut -c1-300
cd $R && python tools/prof_summary.py $O/pmc_nosync $O/pmc_nosync_summary 2>&1 | tail -2; rm -rf $O/pmc_nosync
exit 0
