#!/bin/bash
# round-4 GPU call 7: timing half of call 6 (whose tool invocation failed on a missing default; its parity half passed: 27 tests): the decode step of 1 / 2 / 4
# streams with the split attention's ordered sum inside the o-projection (fa=1) and as its own launch (fa=0), and against the MFMA projections (gv=0)
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4c7; mkdir -p $O
cd $R
timeout -k 10 300 python3 tools/bench_decode_step.py --iters 200 --max-streams 8 \
  b1_nofold:B=1,fa=0 b1_fold:B=1,fa=1 b2_nofold:B=2,fa=0 b2_fold:B=2,fa=1 b4_nofold:B=4,fa=0 b4_fold:B=4,fa=1 b1_nofold2:B=1,fa=0 b1_fold2:B=1,fa=1 \
  b1_fold_p:B=1,fa=1,prof=1 b1_fold_1023:B=1,fa=1,pos=1023 b1_nofold_1023:B=1,fa=0,pos=1023 b1_mfma_1023:B=1,gv=0,pos=1023 b1_mfma:B=1,gv=0 > $O/steps.jsonl 2> $O/steps.err; rc=$?
python3 - <<'PY'
import json,os
for l in open(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r4c7/steps.jsonl"):
    d=json.loads(l); print(d["config"], d["ms_per_step_lane"], d["tokens_per_s_device_time"], d.get("families_us_per_step_lane0",""), d["errors"])
PY
tail -3 $O/steps.err
exit $rc
