#!/bin/bash
# round-2 GPU call 24: loads no longer inside per-element branches (attention_v2 Q operand, GEMM residuals, dec_gemm row tiles);
# gemm_mid K-group count as a knob
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2c24; mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?
tail -5 $O/pytest.log; echo "pytest rc=$rc"
[ $rc -ne 0 ] && exit 1
timeout -k 10 300 python tools/encode_only.py 192 2048 2 > $O/encode_only.json 2> $O/encode_only.err; rc=$?
cut -c1-600 $O/encode_only.json; echo "encode_only rc=$rc"
[ $rc -ne 0 ] && exit 1
timeout -k 10 600 python tools/bench_decode_step.py --max-streams 512 --iters 60 \
  "b512_nw4_prof:B=512,L=1,wm=256,wmode=3,prof=1,nw=4" "b512_nw8_prof:B=512,L=1,wm=256,wmode=3,prof=1,nw=8" "b512_nw0_prof:B=512,L=1,wm=256,wmode=3,prof=1,nw=0" \
  "b512_nw4_l2:B=512,L=2,wm=256,wmode=3,nw=4" "b512_nw8_l2:B=512,L=2,wm=256,wmode=3,nw=8" "b512_nw0_l2:B=512,L=2,wm=256,wmode=3,nw=0" \
  "b128_l3:B=128,L=3" "b1:B=1,L=1" > $O/steps.jsonl 2> $O/steps.err; rc=$?
cut -c1-700 $O/steps.jsonl; echo "microbench rc=$rc"
[ $rc -ne 0 ] && exit 1
timeout -k 10 900 python bench.py --steps 6 --warmup 1 --debug-set mid_nw=4 > $O/bench_nw4.json 2> $O/bench_nw4.err; rc=$?
tail -3 $O/bench_nw4.err | cut -c1-300; cut -c1-400 $O/bench_nw4.json; echo; echo "bench rc=$rc"
exit 0
