#!/bin/bash
# round-2 GPU call 8: parity of the wide-group step, then wide step vs latency step at B = 256..1024 and against lanes
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2c8; mkdir -p $O
cd $R
timeout -k 10 500 python -m pytest tests/test_gpu_round2.py tests/test_gpu_kernels.py -m gpu -x -q -k "wide or gemm or attention" > $O/pytest.log 2>&1; rc=$?
tail -6 $O/pytest.log; echo "pytest rc=$rc"
[ $rc -ne 0 ] && exit 1
show() { python - $1 <<'PY'
import json,sys
for l in open(sys.argv[1]):
    d=json.loads(l); print(f"{d['config']:26s} B={d['B']:4d} L={d['lanes']:2d} skip={d['dbg_skip']} ms/step min={min(d['ms_per_step_lane']):.3f} max={max(d['ms_per_step_lane']):.3f} tok/s={d['tokens_per_s_device_time']} {d['errors'][:1]}")
    if 'families_us_per_step_lane0' in d: print('     ', d['families_us_per_step_lane0'])
PY
}
timeout -k 10 700 python tools/bench_decode_step.py --iters 40 --max-streams 1024 \
  "wide256:B=256,L=1,wm=1" "dec256:B=256,L=1,wm=0" "wide384:B=384,L=1,wm=1" "dec384:B=384,L=1,wm=0" "wide512:B=512,L=1,wm=1" "wide768:B=768,L=1,wm=1" "wide1024:B=1024,L=1,wm=1" \
  "wide128:B=128,L=1,wm=1" "wide768_gemm:B=768,L=1,wm=1,av=1" "wide768_prof:B=768,L=1,wm=1,prof=1" \
  "wide512x2:B=512,L=2,wm=1" "wide384x2:B=384,L=2,wm=1" "wide256x3:B=256,L=3,wm=1" "wide128x3:B=128,L=3,wm=1" \
  > $O/steps.jsonl 2> $O/steps.err; rc=$?
show $O/steps.jsonl; tail -3 $O/steps.err; echo "rc=$rc"
exit 0
