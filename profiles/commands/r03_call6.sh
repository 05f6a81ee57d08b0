#!/bin/bash
# round-3 GPU call 6: RWKV7 state kernel at 4 waves per SIMD (was 1: 256 VGPRs), RWKV loops honour step_sync; rwkv7 PMC passes; bench lines;
# the single-stream line (BASELINE config 2 as the reference CLI runs it)
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3c6; mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_round3.py tests/test_gpu_model.py tests/test_gpu_round2.py tests/test_gpu_fuzz.py -m gpu -x -q -k "rwkv or wide_group" > $O/pytest_rwkv.log 2>&1; rc=$?
tail -4 $O/pytest_rwkv.log; echo "pytest rwkv rc=$rc"
[ $rc -ne 0 ] && exit 1
for shape in "384 2048 128 3" "512 2048 256 2" "1024 2048 1024 1"; do
  timeout -k 10 300 python tools/bench_rwkv7.py $shape >> $O/rwkv7_shapes.jsonl 2>> $O/rwkv7_shapes.err || { echo "shape $shape failed"; tail -3 $O/rwkv7_shapes.err; }
done
cut -c1-330 $O/rwkv7_shapes.jsonl
cd /tmp && export TMPDIR=/tmp
run() { local name=$1; shift; timeout -k 10 900 rocprofv3 "$@" > $O/$name.out 2> $O/$name.err; local rc=$?; echo "$name rc=$rc"; grep "profile_pass\]" $O/$name.err | tail -1; return $rc; }
run kt_rwkv7 --kernel-trace --output-format csv -d $O/kt_rwkv7 -- python3 $R/tools/profile_pass.py rwkv7 && { python3 $R/tools/prof_summary.py $O/kt_rwkv7 $O/r03_passrwkv7_s256_t512 --skip-frac 0.0001; head -12 $O/r03_passrwkv7_s256_t512_kernel_stats.csv; }; rm -rf $O/kt_rwkv7
run pmc_f_rwkv7 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_f_rwkv7 -- python3 $R/tools/profile_pass.py rwkv7 && python3 $R/tools/prof_summary.py $O/pmc_f_rwkv7 $O/r03_pmcrwkv7_FETCH_SIZE; rm -rf $O/pmc_f_rwkv7
run pmc_w_rwkv7 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_w_rwkv7 -- python3 $R/tools/profile_pass.py rwkv7 && python3 $R/tools/prof_summary.py $O/pmc_w_rwkv7 $O/r03_pmcrwkv7_WRITE_SIZE; rm -rf $O/pmc_w_rwkv7
cd $R
timeout -k 10 600 python bench.py --backend rwkv7 --steps 2 --warmup 1 > $O/bench_rwkv7.json 2> $O/bench_rwkv7.err; rc=$?
tail -2 $O/bench_rwkv7.err; cut -c1-300 $O/bench_rwkv7.json; echo "bench rwkv7 rc=$rc"
timeout -k 10 600 python bench.py --steps 8 --warmup 2 > $O/bench_default_k8.json 2> $O/bench_default_k8.err; rc=$?
tail -2 $O/bench_default_k8.err; cut -c1-300 $O/bench_default_k8.json; echo "bench default rc=$rc"
timeout -k 10 600 python bench.py --streams 1 --lanes 1 --lane-streams 1 --sample-tokens 41933 --window 1024 --steps 41 --warmup 0 --cpu-sample 0 --profile-tokens 0 > $O/bench_single_stream.json 2> $O/bench_single_stream.err; rc=$?
tail -1 $O/bench_single_stream.err; python -c "import json;d=json.loads(open('$O/bench_single_stream.json').read().strip().splitlines()[-1]);print('single stream', d['value'], d['compress_tokens_per_s_rank0'], d['decompress_tokens_per_s_rank0'])"
exit 0
