#!/bin/bash
# round-3 GPU call 34: staggered start of the decode-attention blocks (the two blocks of a CU out of phase): attention-only and full steps
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3c34; mkdir -p $O
cd $R
W="wm=256,wmode=3"
CFG=""
for stg in 0 8 16 32 528 544 0; do CFG="$CFG attn_s$stg:B=512,L=1,av=2,stg=$stg,$W"; done
for stg in 0 16 32 544 0; do CFG="$CFG both_s$stg:B=512,L=1,stg=$stg,$W both2l_s$stg:B=512,L=2,stg=$stg,$W"; done
timeout -k 10 500 python tools/bench_decode_step.py --iters 40 --max-streams 512 --sym 0 $CFG > $O/steps.jsonl 2> $O/steps.err || { tail -3 $O/steps.err; exit 1; }
python -c "
import json
for l in open('$O/steps.jsonl'):
    d=json.loads(l); print(d['config'], d['ms_per_step_lane'])"
