#!/bin/bash
# round-3 GPU call 4: full suite after the B = 1 latency work (loads hoisted in dec_gemm_kernel, 64 loads in flight in the small-grid attention),
# RWKV7 wide step on grouped K-group GEMMs + state kernel with the tile load first; B = 1 step time and kernel trace; RWKV lane shapes;
# rocprofv3 --pmc (600 tokens) without --step-sync
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3c4; mkdir -p $O
cd $R
timeout -k 10 1200 python -m pytest tests -m gpu -x -q --durations=5 > $O/pytest_all.log 2>&1; rc=$?
tail -12 $O/pytest_all.log; echo "pytest all rc=$rc"
[ $rc -ne 0 ] && exit 1
timeout -k 10 300 python tools/bench_decode_step.py --iters 200 --max-streams 16 "b1:B=1,L=1,g=1" "b8:B=8,L=1,g=1" "b16:B=16,L=1,g=1" "b1_prof:B=1,L=1,g=0,prof=1" > $O/b1_steps.jsonl 2> $O/b1_steps.err; rc=$?
cut -c1-700 $O/b1_steps.jsonl; echo "b1 steps rc=$rc"
for shape in "384 2048 128 3" "512 2048 256 2" "512 2048 512 1" "1024 2048 1024 1"; do
  timeout -k 10 300 python tools/bench_rwkv7.py $shape >> $O/rwkv7_shapes.jsonl 2>> $O/rwkv7_shapes.err || { echo "shape $shape failed"; tail -3 $O/rwkv7_shapes.err; }
done
cut -c1-330 $O/rwkv7_shapes.jsonl
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/b1_trace -- python3 $R/tools/bench_decode_step.py --iters 20 --max-streams 1 "b1:B=1,L=1,g=1" > $O/b1_trace.out 2> $O/b1_trace.err; rc=$?
echo "b1 trace rc=$rc"
python3 $R/tools/trace_gaps.py $O/b1_trace --last 1550 > $O/b1_gaps.txt 2>&1; head -14 $O/b1_gaps.txt; rm -rf $O/b1_trace
timeout -k 10 500 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_nosync -- python3 $R/bench.py --streams 128 --lanes 1 --lane-streams 128 --no-graph --no-events --overlap 0 --sample-tokens 600 --window 600 --steps 1 --warmup 0 --cpu-sample 0 --profile-tokens 0 > $O/pmc_nosync.out 2> $O/pmc_nosync.err; rc=$?
echo "pmc 600 tokens (no --step-sync, library bounds its queue depth) rc=$rc"; grep -v "^    @" $O/pmc_nosync.err | tail -3 | cut -c1-300; tail -1 $O/pmc_nosync.out | cut -c1-200
cd $R && python tools/prof_summary.py $O/pmc_nosync $O/pmc_nosync_600 2>&1 | tail -2; rm -rf $O/pmc_nosync
exit 0
