#!/bin/bash
# round-4 GPU call 42: gemm_mid_kernel with buffer-descriptor addressing (as the big-M GEMM): wide-group parity tests, then the 512-stream wide step A/B against the build before it
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4c42; mkdir -p $O
cd $R
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q -k "wide or mid or threshold or lanes or round2 or round3" > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
for rep in 1 2; do
  for lib in libcandlezip_hip_base.so libcandlezip_hip.so; do
    CANDLEZIP_AMD_LIB=$R/candlezip_amd/lib/$lib timeout -k 10 200 python3 tools/bench_decode_step.py --iters 100 --max-streams 512 \
      $lib.chain767:B=512,wm=256,wmode=3,av=1 $lib.full767:B=512,wm=256,wmode=3 $lib.two767:B=512,L=2,wm=256,wmode=3 >> $O/steps.jsonl 2>> $O/steps.err || exit 1
  done
done
python3 - <<'PY'
import json,os
for l in open(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r4c42/steps.jsonl"):
    d=json.loads(l); print(d["config"], d["ms_per_step_lane"], d["errors"])
PY
