#!/bin/bash
# round-3 GPU call 16: gemm_mid K groups per block: 5 (K = 576 in two rounds), 6, 5-and-6 by K against 4
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3c16; mkdir -p $O
cd $R
timeout -k 10 400 python tools/bench_decode_step.py --iters 60 --max-streams 512 "nw4:B=512,L=1,wm=256,wmode=3,nw=4" "nw5:B=512,L=1,wm=256,wmode=3,nw=5" "nw6:B=512,L=1,wm=256,wmode=3,nw=6" "nw56:B=512,L=1,wm=256,wmode=3,nw=56" "nw4_2l:B=512,L=2,wm=256,wmode=3,nw=4" "nw56_2l:B=512,L=2,wm=256,wmode=3,nw=56" "nw5_2l:B=512,L=2,wm=256,wmode=3,nw=5" > $O/steps_nw.jsonl 2> $O/steps_nw.err; cut -c1-250 $O/steps_nw.jsonl; tail -2 $O/steps_nw.err
exit 0
