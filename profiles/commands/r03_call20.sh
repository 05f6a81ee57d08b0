#!/bin/bash
# round-3 GPU call 20: lane-shape sweep of the wide decode step at the final kernels (more concurrent lanes: does attention of one hide the GEMM chain of another?)
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3c20; mkdir -p $O
cd $R
timeout -k 10 400 python tools/bench_decode_step.py --iters 40 --max-streams 512 "b512_2l:B=512,L=2,wm=256,wmode=3" "b512_3l:B=512,L=3,wm=256,wmode=3" "b512_4l:B=512,L=4,wm=256,wmode=3" "b384_3l:B=384,L=3,wm=256,wmode=3" "b384_4l:B=384,L=4,wm=256,wmode=3" > $O/steps.jsonl 2> $O/steps.err; rc=$?; cut -c1-260 $O/steps.jsonl; tail -3 $O/steps.err
[ $rc -ne 0 ] && exit $rc
timeout -k 10 400 python tools/bench_decode_step.py --iters 40 --max-streams 256 "b256_4l:B=256,L=4,wm=256,wmode=3" "b256_6l:B=256,L=6,wm=256,wmode=3" > $O/steps256.jsonl 2> $O/steps256.err; rc=$?; cut -c1-260 $O/steps256.jsonl; tail -3 $O/steps256.err
[ $rc -ne 0 ] && exit $rc
timeout -k 10 400 python tools/bench_decode_step.py --iters 40 --max-streams 1024 "b1024_2l:B=1024,L=2,wm=256,wmode=3" "b768_2l:B=768,L=2,wm=256,wmode=3" > $O/steps1024.jsonl 2> $O/steps1024.err; rc=$?; cut -c1-260 $O/steps1024.jsonl; tail -3 $O/steps1024.err
exit $rc
