#!/bin/bash
# round-4 GPU call 74: four-wave decode-attention blocks at THREE blocks per CU (LDS pad: 1 536 blocks on 768 slots = two whole generations) against four
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4c74; mkdir -p $O
cd $R
for rep in 1 2; do
  timeout -k 10 300 python3 tools/bench_decode_step.py --iters 100 --max-streams 512 \
    w4.att767:B=512,wm=256,wmode=3,av=2,aw=4 w4p.att767:B=512,wm=256,wmode=3,av=2,aw=4,pad=20480 \
    w4.att1023:B=512,wm=256,wmode=3,av=2,pos=1023,aw=4 w4p.att1023:B=512,wm=256,wmode=3,av=2,pos=1023,aw=4,pad=20480 \
    w4.two767:B=512,L=2,wm=256,wmode=3,aw=4 w4p.two767:B=512,L=2,wm=256,wmode=3,aw=4,pad=20480 >> $O/steps.jsonl 2>> $O/steps.err || { tail -5 $O/steps.err; exit 1; }
done
python3 - <<'PY'
import json,os
for l in open(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r4c74/steps.jsonl"):
    d=json.loads(l); print(d["config"], d["ms_per_step_lane"], d["errors"])
PY
