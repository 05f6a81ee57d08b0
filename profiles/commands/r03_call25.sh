#!/bin/bash
# round-3 GPU call 25: whole-device decode attention at three blocks per CU (80 VGPRs, 76 B scratch) against two (106 VGPRs): attention-only and full steps
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3c25; mkdir -p $O
cd $R
W="wm=256,wmode=3"
CFG="attn_1l:B=512,L=1,av=2,$W attn_1l_1023:B=512,L=1,av=2,pos=1023,$W both_1l:B=512,L=1,$W both_2l:B=512,L=2,$W"
timeout -k 10 300 python tools/bench_decode_step.py --iters 40 --max-streams 512 --sym 0 $CFG > $O/steps_occ4.jsonl 2> $O/steps_occ4.err || exit 1
cut -c1-260 $O/steps_occ4.jsonl
export CANDLEZIP_AMD_LIB=$R/candlezip_amd/lib/libcandlezip_hip_daocc6.so
timeout -k 10 300 python tools/bench_decode_step.py --iters 40 --max-streams 512 --sym 0 $CFG > $O/steps_occ6.jsonl 2> $O/steps_occ6.err || exit 1
cut -c1-260 $O/steps_occ6.jsonl
