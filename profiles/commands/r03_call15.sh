#!/bin/bash
# round-3 GPU call 15: quad epilogue with one wave per 64-chunk (nine / twelve waves, mid_nw = 0) and eight K groups against four; parity of those forms
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3c15; mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_round2.py -m gpu -x -q -k "wide_group" > $O/pytest.log 2>&1; rc=$?
tail -4 $O/pytest.log; echo "pytest rc=$rc"
[ $rc -ne 0 ] && exit 1
timeout -k 10 400 python tools/bench_decode_step.py --iters 60 --max-streams 512 "nw4:B=512,L=1,wm=256,wmode=3,nw=4" "nw0:B=512,L=1,wm=256,wmode=3,nw=0" "nw8:B=512,L=1,wm=256,wmode=3,nw=8" "nw4_2l:B=512,L=2,wm=256,wmode=3,nw=4" "nw0_2l:B=512,L=2,wm=256,wmode=3,nw=0" > $O/steps_nw.jsonl 2> $O/steps_nw.err; cut -c1-250 $O/steps_nw.jsonl
exit 0
