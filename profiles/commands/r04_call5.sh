#!/bin/bash
# round-4 GPU call 5: the VALU projections of narrow groups (kernels_gemv.hip): parity (every path of the step-form test, both shapes; the single-stream
# tests of the model suite), then the decode step of 1 / 2 / 4 streams at 767 keys with and without them (graph replay; eager with per-family events)
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4c5; mkdir -p $O
cd $R && timeout -k 10 800 python3 -m pytest tests/test_gpu_round2.py tests/test_gpu_model.py tests/test_gpu_round3.py -m gpu -x -q -k "wide_group or trait or smollm2_135m or config5 or split_decode or past_the_initial" > $O/pytest.log 2>&1; rc=$?
tail -3 $O/pytest.log
[ $rc -ne 0 ] && exit 1
timeout -k 10 300 python3 tools/bench_decode_step.py --iters 200 --max-streams 8 \
  b1_mfma:B=1,gv=0 b1_gemv:B=1,gv=4 b2_mfma:B=2,gv=0 b2_gemv:B=2,gv=4 b4_mfma:B=4,gv=0 b4_gemv:B=4,gv=4 b1_mfma2:B=1,gv=0 b1_gemv2:B=1,gv=4 \
  b1_mfma_p:B=1,gv=0,prof=1 b1_gemv_p:B=1,gv=4,prof=1 b4_mfma_p:B=4,gv=0,prof=1 b4_gemv_p:B=4,gv=4,prof=1 > $O/steps.jsonl 2> $O/steps.err; rc=$?
python3 - <<'PY'
import json,os
for l in open(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r4c5/steps.jsonl"):
    d=json.loads(l); print(d["config"], d["ms_per_step_lane"], d["tokens_per_s_device_time"], d.get("families_us_per_step_lane0",""), d["errors"])
PY
tail -3 $O/steps.err
exit $rc
