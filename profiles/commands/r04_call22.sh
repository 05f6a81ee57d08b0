#!/bin/bash
# round-4 GPU call 22: where the step of 24 .. 192 streams (the narrow MFMA form) spends its time: per-family events, and the wide form (wm) from 128 streams on
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4c22; mkdir -p $O
cd $R
timeout -k 10 400 python3 tools/bench_decode_step.py --iters 60 --max-streams 256 \
  b24:B=24,prof=1 b48:B=48,prof=1 b96:B=96,prof=1 b192:B=192,prof=1 b24g:B=24 b48g:B=48 b96g:B=96 b128g:B=128 b192g:B=192 b256g:B=256 \
  b128w:B=128,wm=128,wmode=3 b192w:B=192,wm=128,wmode=3 b96w:B=96,wm=64,wmode=3 > $O/steps.jsonl 2> $O/steps.err; rc=$?
python3 - <<'PY'
import json,os
for l in open(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r4c22/steps.jsonl"):
    d=json.loads(l); print(d["config"], d["ms_per_step_lane"], d["tokens_per_s_device_time"], d.get("families_us_per_step_lane0",""), d["errors"])
PY
tail -3 $O/steps.err
exit $rc
