#!/bin/bash
# round-4 GPU call 77: decode attention capped at 96 VGPRs (five waves per SIMD: room for a projection block of the other lane beside four attention waves; 18 / 35 spilled registers): two lanes, attention alone
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4c77; mkdir -p $O
cd $R
for lib in libcandlezip_hip.so libcandlezip_hip_daocc5.so; do
  CANDLEZIP_AMD_LIB=$R/candlezip_amd/lib/$lib timeout -k 10 300 python3 tools/bench_decode_step.py --iters 100 --max-streams 512 \
    $lib.w8.att767:B=512,wm=256,wmode=3,av=2 $lib.w4.att767:B=512,wm=256,wmode=3,av=2,aw=4 \
    $lib.w8.two767:B=512,L=2,wm=256,wmode=3 $lib.w4.two767:B=512,L=2,wm=256,wmode=3,aw=4 >> $O/steps.jsonl 2>> $O/steps.err || { tail -5 $O/steps.err; exit 1; }
done
python3 - <<'PY'
import json,os
for l in open(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r4c77/steps.jsonl"):
    d=json.loads(l); print(d["config"], d["ms_per_step_lane"], d["errors"])
PY
