#!/bin/bash
# round-2 GPU call 37: one decode-attention block per CU (LDS padding) so that the other lane's GEMM blocks fit beside it
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2c37; mkdir -p $O
cd $R
timeout -k 10 500 python tools/bench_decode_step.py --max-streams 512 --iters 100 \
  "b512_l2_pad0:B=512,L=2,wm=256,wmode=3" "b512_l2_pad30k:B=512,L=2,wm=256,wmode=3,pad=30720" "b512_l2_pad56k:B=512,L=2,wm=256,wmode=3,pad=57344" \
  "b512_l1_pad0:B=512,L=1,wm=256,wmode=3" "b512_l1_pad56k:B=512,L=1,wm=256,wmode=3,pad=57344" \
  "b341_l3_pad0:B=341,L=3,wm=256,wmode=3" "b341_l3_pad56k:B=341,L=3,wm=256,wmode=3,pad=57344" > $O/steps.jsonl 2> $O/steps.err; rc=$?
python - <<'PY'
import json
for l in open("gpurun_out/r2c37/steps.jsonl"):
    d=json.loads(l); print(d["config"], d["ms_per_step_lane"], d["tokens_per_s_device_time"], d["errors"])
PY
echo "microbench rc=$rc"; tail -3 $O/steps.err
exit 0
