#!/bin/bash
# round-2 GPU call 4: lanes x streams-per-lane sweep of the decode step, attention-only / GEMM-only concurrency, attention_v2 parity + encode timing
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2c4; mkdir -p $O
cd $R
timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -m gpu -x -q > $O/pytest.log 2>&1; rc=$?
tail -4 $O/pytest.log; echo "pytest rc=$rc"
[ $rc -ge 124 ] && exit 1
timeout -k 10 600 python tools/bench_decode_step.py --iters 50 --max-streams 256 \
  "g3_b128:B=128,L=3" "g4_b96:B=96,L=4" "g4_b85:B=85,L=4" "g5_b85:B=85,L=5" "g6_b64:B=64,L=6" "g8_b64:B=64,L=8" "g8_b48:B=48,L=8" "g10_b48:B=48,L=10" "g12_b32:B=32,L=12" \
  "g2_b256:B=256,L=2" "g2_b192:B=192,L=2" "g1_b85:B=85,L=1" "g1_b256:B=256,L=1" \
  "attn_only_g1:B=128,L=1,av=2" "attn_only_g3:B=128,L=3,av=2" "attn_only_g8b48:B=48,L=8,av=2" "attn_only_g1b85:B=85,L=1,av=2" "attn_only_g1b256:B=256,L=1,av=2" \
  "gemm_only_g1:B=128,L=1,av=1" "gemm_only_g3:B=128,L=3,av=1" "gemm_only_g8b48:B=48,L=8,av=1" "gemm_only_g1b48:B=48,L=1,av=1" \
  > $O/steps.jsonl 2> $O/steps.err; rc=$?
python - <<'PY'
import json
for l in open('gpurun_out/r2c4/steps.jsonl'):
    d=json.loads(l); print(f"{d['config']:18s} B={d['B']:3d} L={d['lanes']:2d} skip={d['dbg_skip']} ms/step(max)={max(d['ms_per_step_lane']):.3f} tok/s={d['tokens_per_s_device_time']} {d['errors'][:1]}")
PY
tail -3 $O/steps.err; echo "microbench rc=$rc"
exit 0
