#!/bin/bash
# round-3 GPU call 26: transposed K of the decode cache in 64-key tiles (16 KB contiguous per chunk): attention-only / full step times, then the whole GPU suite
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3c26; mkdir -p $O
cd $R
W="wm=256,wmode=3"
timeout -k 10 300 python tools/bench_decode_step.py --iters 40 --max-streams 512 --sym 0 "attn_1l:B=512,L=1,av=2,$W" "attn_1l_1023:B=512,L=1,av=2,pos=1023,$W" "both_1l:B=512,L=1,$W" "both_2l:B=512,L=2,$W" "b1:B=1,L=1" "b8:B=8,L=1" "b128:B=128,L=1" > $O/steps.jsonl 2> $O/steps.err || { tail -5 $O/steps.err; exit 1; }
cut -c1-250 $O/steps.jsonl
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?
tail -5 $O/pytest.log; echo "pytest rc=$rc"
exit $rc
