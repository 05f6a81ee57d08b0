#!/bin/bash
# round-2 GPU call 23: gemm_mid with 9 or 12 K groups (one or two rounds), RoPE in the QKV epilogue, 4-row RMSNorm
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2c23; mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?
tail -5 $O/pytest.log; echo "pytest rc=$rc"
[ $rc -ne 0 ] && exit 1
timeout -k 10 600 python tools/bench_decode_step.py --max-streams 512 --iters 60 \
  "b512_prof:B=512,L=1,wm=256,wmode=3,prof=1" "b512_l1:B=512,L=1,wm=256,wmode=3" "b512_l2:B=512,L=2,wm=256,wmode=3" \
  "b384_prof:B=384,L=1,wm=256,wmode=3,prof=1" "b384_l2:B=384,L=2,wm=256,wmode=3" > $O/steps.jsonl 2> $O/steps.err; rc=$?
cut -c1-700 $O/steps.jsonl; echo "microbench rc=$rc"
[ $rc -ne 0 ] && exit 1
timeout -k 10 900 python bench.py --steps 6 --warmup 1 > $O/bench_k6.json 2> $O/bench_k6.err; rc=$?
tail -4 $O/bench_k6.err | cut -c1-300; cut -c1-400 $O/bench_k6.json; echo; echo "bench rc=$rc"
exit 0
