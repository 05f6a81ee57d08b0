#!/bin/bash
# round-4 GPU call 68: decode attention with FOUR waves per block (each wave walks up to 8 chunks; four blocks per CU) against eight: parity tests of the wide / narrow decodes with the knob, attention-only and full steps at 511 / 767 / 1023 keys
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4c68; mkdir -p $O
cd $R
for rep in 1 2; do
  timeout -k 10 300 python3 tools/bench_decode_step.py --iters 100 --max-streams 512 \
    w8.att511:B=512,wm=256,wmode=3,av=2,pos=511 w4.att511:B=512,wm=256,wmode=3,av=2,pos=511,aw=4 \
    w8.att767:B=512,wm=256,wmode=3,av=2 w4.att767:B=512,wm=256,wmode=3,av=2,aw=4 \
    w8.att1023:B=512,wm=256,wmode=3,av=2,pos=1023 w4.att1023:B=512,wm=256,wmode=3,av=2,pos=1023,aw=4 \
    w8.full767:B=512,wm=256,wmode=3 w4.full767:B=512,wm=256,wmode=3,aw=4 w8.two767:B=512,L=2,wm=256,wmode=3 w4.two767:B=512,L=2,wm=256,wmode=3,aw=4 >> $O/steps.jsonl 2>> $O/steps.err || { tail -5 $O/steps.err; exit 1; }
done
python3 - <<'PY'
import json,os
for l in open(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r4c68/steps.jsonl"):
    d=json.loads(l); print(d["config"], d["ms_per_step_lane"], d["errors"])
PY
