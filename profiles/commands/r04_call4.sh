#!/bin/bash
# round-4 GPU call 4: gate|up with the K-group assignment of a block's waves turned per block (mid_rot): per-family events of the wide step at 512 streams
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4c4; mkdir -p $O
cd $R
timeout -k 10 300 python3 tools/bench_decode_step.py --iters 60 --max-streams 512 \
  rotn:B=512,wm=256,wmode=3,rot=-1,prof=1 rot8:B=512,wm=256,wmode=3,rot=8,prof=1 rot0:B=512,wm=256,wmode=3,rot=0,prof=1 rot3:B=512,wm=256,wmode=3,rot=3,prof=1 rot2:B=512,wm=256,wmode=3,rot=2,prof=1 \
  rotn_g:B=512,wm=256,wmode=3,rot=-1 rot8_g:B=512,wm=256,wmode=3,rot=8 rot0_g:B=512,wm=256,wmode=3,rot=0 old_g:B=512,wm=256,wmode=3,cpw=0 > $O/steps.jsonl 2> $O/steps.err; rc=$?
python3 - <<'PY'
import json,os
for l in open(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r4c4/steps.jsonl"):
    d=json.loads(l); print(d["config"], d["ms_per_step_lane"], d.get("families_us_per_step_lane0",{}).get("dec_gemm_swiglu"), d["errors"])
PY
tail -3 $O/steps.err
exit $rc
