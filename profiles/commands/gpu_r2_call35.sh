#!/bin/bash
# round-2 GPU call 35: where does the wide (gemm_mid) step start to beat the narrow (dec_gemm) step?
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2c35; mkdir -p $O
cd $R
timeout -k 10 500 python tools/bench_decode_step.py --max-streams 256 --iters 100 \
  "b64_narrow:B=64,L=1,wm=1024" "b64_wide:B=64,L=1,wm=1,wmode=3" \
  "b96_narrow:B=96,L=1,wm=1024" "b96_wide:B=96,L=1,wm=1,wmode=3" \
  "b128_narrow:B=128,L=1,wm=1024" "b128_wide:B=128,L=1,wm=1,wmode=3" \
  "b192_narrow:B=192,L=1,wm=1024" "b192_wide:B=192,L=1,wm=1,wmode=3" \
  "b256_narrow:B=256,L=1,wm=1024" "b256_wide:B=256,L=1,wm=1,wmode=3" \
  "b96x2_narrow:B=96,L=2,wm=1024" "b96x2_wide:B=96,L=2,wm=1,wmode=3" \
  "b128x3_narrow:B=128,L=3,wm=1024" "b128x3_wide:B=128,L=3,wm=1,wmode=3" > $O/steps.jsonl 2> $O/steps.err; rc=$?
python - <<'PY'
import json
for l in open("gpurun_out/r2c35/steps.jsonl"):
    d=json.loads(l); print(d["config"], d["ms_per_step_lane"], d["tokens_per_s_device_time"], d["errors"])
PY
echo "microbench rc=$rc"
exit 0
