#!/bin/bash
# round-4 GPU call 29: decode attention with K streamed in two rounds of 32 d rows per chunk (make da_kround32) against the shipped four rounds of 16: the 512-stream step,
# attention alone and whole, alternating builds
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4c29; mkdir -p $O
cd $R
for rep in 1 2; do
  for lib in libcandlezip_hip.so libcandlezip_hip_dakround32.so; do
    CANDLEZIP_AMD_LIB=$R/candlezip_amd/lib/$lib timeout -k 10 200 python3 tools/bench_decode_step.py --iters 100 --max-streams 512 \
      $lib.att767:B=512,wm=256,wmode=3,av=2 $lib.full767:B=512,wm=256,wmode=3 $lib.att1023:B=512,wm=256,wmode=3,av=2,pos=1023 $lib.full1023:B=512,wm=256,wmode=3,pos=1023 >> $O/steps.jsonl 2>> $O/steps.err || exit 1
  done
done
python3 - <<'PY'
import json,os
for l in open(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r4c29/steps.jsonl"):
    d=json.loads(l); print(d["config"], d["ms_per_step_lane"], d["errors"])
PY
