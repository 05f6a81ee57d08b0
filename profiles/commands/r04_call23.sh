#!/bin/bash
# round-4 GPU call 23: the wide form's threshold (128 .. 224 streams), and the split attention for 24 / 32 streams
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4c23; mkdir -p $O
cd $R
timeout -k 10 400 python3 tools/bench_decode_step.py --iters 60 --max-streams 256 \
  b144n:B=144 b144w:B=144,wm=128,wmode=3 b160n:B=160 b160w:B=160,wm=128,wmode=3 b176n:B=176 b176w:B=176,wm=128,wmode=3 b224n:B=224 b224w:B=224,wm=128,wmode=3 b256w:B=256,wm=128,wmode=3 \
  b24_sp36:B=24,sp=36 b24_spall:B=24,sp=100000 b32_sp36:B=32,sp=36 b32_spall:B=32,sp=100000 b20_sp36:B=20,sp=36 b20_spall:B=20,sp=100000 > $O/steps.jsonl 2> $O/steps.err; rc=$?
python3 - <<'PY'
import json,os
for l in open(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r4c23/steps.jsonl"):
    d=json.loads(l); print(d["config"], d["ms_per_step_lane"], d["tokens_per_s_device_time"], d["errors"])
PY
tail -3 $O/steps.err
exit $rc
