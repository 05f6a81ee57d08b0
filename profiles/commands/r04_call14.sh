#!/bin/bash
# round-4 GPU call 14: the split attention (its ordered sum inside the o-projection) for groups of 6..16 streams: dec_attn_split_max 36 (B <= 4) against 144 (B <= 16), VALU projections
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4c14; mkdir -p $O
cd $R
timeout -k 10 400 python3 tools/bench_decode_step.py --iters 100 --max-streams 64 \
  b6_sp36:B=6,gv=16,sp=36 b6_sp144:B=6,gv=16,sp=144 b8_sp36:B=8,gv=16,sp=36 b8_sp144:B=8,gv=16,sp=144 b12_sp36:B=12,gv=16,sp=36 b12_sp144:B=12,gv=16,sp=144 \
  b16_sp36:B=16,gv=16,sp=36 b16_sp144:B=16,gv=16,sp=144 b8_sp144_1023:B=8,gv=16,sp=144,pos=1023 b8_sp36_1023:B=8,gv=16,sp=36,pos=1023 > $O/steps.jsonl 2> $O/steps.err; rc=$?
python3 - <<'PY'
import json,os
for l in open(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r4c14/steps.jsonl"):
    d=json.loads(l); print(d["config"], d["ms_per_step_lane"], d["tokens_per_s_device_time"], d["errors"])
PY
tail -3 $O/steps.err
exit $rc
