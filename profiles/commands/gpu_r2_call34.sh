#!/bin/bash
# round-2 GPU call 34: step-time check after dropping the dec_attn latency mode (wide groups must not have moved)
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2c34; mkdir -p $O
cd $R
timeout -k 10 300 python tools/bench_decode_step.py --max-streams 512 --iters 100 "b1:B=1,L=1" "b512_l2:B=512,L=2,wm=256,wmode=3" "b512_l2_again:B=512,L=2,wm=256,wmode=3" "b512_prof:B=512,L=1,wm=256,wmode=3,prof=1" > $O/steps.jsonl 2> $O/steps.err; rc=$?
cut -c1-800 $O/steps.jsonl; echo "microbench rc=$rc"
exit 0
