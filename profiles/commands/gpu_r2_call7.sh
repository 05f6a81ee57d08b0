#!/bin/bash
# round-2 GPU call 7: ONE wide lock-step group (B up to 1024) against lanes
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2c7; mkdir -p $O
cd $R
show() { python - $1 <<'PY'
import json,sys
for l in open(sys.argv[1]):
    d=json.loads(l); print(f"{d['config']:26s} B={d['B']:4d} L={d['lanes']:2d} skip={d['dbg_skip']} ms/step min={min(d['ms_per_step_lane']):.3f} max={max(d['ms_per_step_lane']):.3f} tok/s={d['tokens_per_s_device_time']} {d['errors'][:1]}")
    if 'families_us_per_step_lane0' in d: print('     ', d['families_us_per_step_lane0'])
PY
}
timeout -k 10 700 python tools/bench_decode_step.py --iters 40 --max-streams 1024 \
  "w256:B=256,L=1" "w384:B=384,L=1" "w512:B=512,L=1" "w768:B=768,L=1" "w1024:B=1024,L=1" \
  "w768_gt768:B=768,L=1,gt=768" "w768_gt1536:B=768,L=1,gt=1536" "w768_gt192:B=768,L=1,gt=192" \
  "w768_attn:B=768,L=1,av=2" "w768_gemm:B=768,L=1,av=1" "w768_prof:B=768,L=1,prof=1" \
  "w512x2:B=512,L=2" "w384x2:B=384,L=2" \
  > $O/steps.jsonl 2> $O/steps.err; rc=$?
show $O/steps.jsonl; tail -3 $O/steps.err; echo "rc=$rc"
exit 0
