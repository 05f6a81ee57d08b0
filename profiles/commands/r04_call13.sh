#!/bin/bash
# round-4 GPU call 13: do the VALU projections (row groups of four per block) pay beyond four streams?  the step of 8 / 16 / 32 / 64 streams at 767 keys,
# MFMA projections (gv=0) against VALU (gv=64)
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4c13; mkdir -p $O
cd $R
timeout -k 10 400 python3 tools/bench_decode_step.py --iters 100 --max-streams 64 \
  b8_mfma:B=8,gv=0 b8_valu:B=8,gv=64 b16_mfma:B=16,gv=0 b16_valu:B=16,gv=64 b32_mfma:B=32,gv=0 b32_valu:B=32,gv=64 b64_mfma:B=64,gv=0 b64_valu:B=64,gv=64 \
  b6_mfma:B=6,gv=0 b6_valu:B=6,gv=64 b12_mfma:B=12,gv=0 b12_valu:B=12,gv=64 > $O/steps.jsonl 2> $O/steps.err; rc=$?
python3 - <<'PY'
import json,os
for l in open(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r4c13/steps.jsonl"):
    d=json.loads(l); print(d["config"], d["ms_per_step_lane"], d["tokens_per_s_device_time"], d["errors"])
PY
tail -3 $O/steps.err
exit $rc
