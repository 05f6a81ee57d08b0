#!/bin/bash
# round-2 GPU call 33: dec_attn latency mode (all K loads of a chunk at once, V slab fetched before the softmax) for one block per query head
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2c33; mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?
tail -4 $O/pytest.log; echo "pytest rc=$rc"
[ $rc -ne 0 ] && exit 1
timeout -k 10 300 python tools/bench_decode_step.py --max-streams 512 --iters 200 "b1_prof:B=1,L=1,prof=1" "b1:B=1,L=1" "b4:B=4,L=1" "b8:B=8,L=1" "b14:B=14,L=1" "b512_l2:B=512,L=2,wm=256,wmode=3" > $O/steps.jsonl 2> $O/steps.err; rc=$?
cut -c1-800 $O/steps.jsonl; echo "microbench rc=$rc"
exit 0
