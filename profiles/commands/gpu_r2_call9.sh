#!/bin/bash
# round-2 GPU call 9: parity of dec_gemm's 32 x 64 blocks; wide groups: tiled-GEMM step vs dec_gemm (old shape / 32 x 64 blocks)
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2c9; mkdir -p $O
cd $R
timeout -k 10 500 python -m pytest tests/test_gpu_round2.py -m gpu -x -q -k "wide" > $O/pytest.log 2>&1; rc=$?
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
  "np2_256:B=256,L=1,wm=0" "old_256:B=256,L=1,wm=0,dw=100000" "np2_384:B=384,L=1,wm=0" "np2_512:B=512,L=1,wm=0" "np2_768:B=768,L=1,wm=0" "old_768:B=768,L=1,wm=0,dw=100000" "tiled_768:B=768,L=1,wm=1" "np2_1024:B=1024,L=1,wm=0" \
  "np2_768_gemm:B=768,L=1,wm=0,av=1" "np2_768_prof:B=768,L=1,wm=0,prof=1" \
  "np2_512x2:B=512,L=2,wm=0" "np2_384x2:B=384,L=2,wm=0" "np2_256x3:B=256,L=3,wm=0" \
  > $O/steps.jsonl 2> $O/steps.err; rc=$?
show $O/steps.jsonl; tail -3 $O/steps.err; echo "rc=$rc"
exit 0
