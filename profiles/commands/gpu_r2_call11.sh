#!/bin/bash
# round-2 GPU call 11: parity of the hybrid / mid-GEMM wide steps; their speed at B = 512 / 768 (attention LDS now sized to the run)
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2c11; mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_round2.py -m gpu -x -q -k "wide or sliced" > $O/pytest.log 2>&1; rc=$?
tail -8 $O/pytest.log; echo "pytest rc=$rc"
show() { python - $1 <<'PY'
import json,sys
for l in open(sys.argv[1]):
    d=json.loads(l); print(f"{d['config']:26s} B={d['B']:4d} L={d['lanes']:2d} skip={d['dbg_skip']} ms/step min={min(d['ms_per_step_lane']):.3f} max={max(d['ms_per_step_lane']):.3f} tok/s={d['tokens_per_s_device_time']} {d['errors'][:1]}")
    if 'families_us_per_step_lane0' in d: print('     ', d['families_us_per_step_lane0'])
PY
}
timeout -k 10 700 python tools/bench_decode_step.py --iters 40 --max-streams 1024 \
  "tiled_768:B=768,L=1,wm=1,wmode=1" "hybrid_768:B=768,L=1,wm=1,wmode=2" "mid_768:B=768,L=1,wm=1,wmode=3" "mid_512:B=512,L=1,wm=1,wmode=3" "mid_1024:B=1024,L=1,wm=1,wmode=3" "mid_256:B=256,L=1,wm=1,wmode=3" \
  "mid_768_gemm:B=768,L=1,wm=1,wmode=3,av=1" "mid_768_prof:B=768,L=1,wm=1,wmode=3,prof=1" "hybrid_768_prof:B=768,L=1,wm=1,wmode=2,prof=1" \
  "mid_512x2:B=512,L=2,wm=1,wmode=3" "tiled_512x2:B=512,L=2,wm=1,wmode=1" "mid_384x2:B=384,L=2,wm=1,wmode=3" "mid_256x3:B=256,L=3,wm=1,wmode=3" "dec_128x3:B=128,L=3,wm=0" \
  > $O/steps.jsonl 2> $O/steps.err; rc=$?
show $O/steps.jsonl; tail -3 $O/steps.err; echo "rc=$rc"
exit 0
