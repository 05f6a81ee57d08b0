#!/bin/bash
# round-4 GPU call 6: the split attention's ordered sum folded into the o-projection (gemv_fold_attn): parity, then the step of 1 / 2 / 4 streams with and without
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4c6; mkdir -p $O
rc=0

[ $rc -ne 0 ] && exit 1
timeout -k 10 300 python3 tools/bench_decode_step.py --iters 200 --max-streams 8 \
  b1_nofold:B=1,fa=0 b1_fold:B=1,fa=1 b2_nofold:B=2,fa=0 b2_fold:B=2,fa=1 b4_nofold:B=4,fa=0 b4_fold:B=4,fa=1 b1_nofold2:B=1,fa=0 b1_fold2:B=1,fa=1 \
  b1_fold_p:B=1,fa=1,prof=1 b1_fold_1023:B=1,fa=1,pos=1023 b1_nofold_1023:B=1,fa=0,pos=1023 b1_mfma_1023:B=1,gv=0,pos=1023 > $O/steps.jsonl 2> $O/steps.err; rc=$?
python3 - <<'PY'
import json,os
for l in open(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r4c6/steps.jsonl"):
    d=json.loads(l); print(d["config"], d["ms_per_step_lane"], d["tokens_per_s_device_time"], d.get("families_us_per_step_lane0",""), d["errors"])
PY
tail -3 $O/steps.err
exit $rc
