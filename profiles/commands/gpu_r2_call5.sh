#!/bin/bash
# round-2 GPU call 5: lane sweep again with the lanes' timed loops started together; then the real pipeline (re-primes included) for four lane shapes
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2c5; mkdir -p $O
cd $R
timeout -k 10 600 python tools/bench_decode_step.py --iters 120 --max-streams 128 \
  "g1_b128:B=128,L=1" "g3_b128:B=128,L=3" "g4_b96:B=96,L=4" "g6_b64:B=64,L=6" "g8_b48:B=48,L=8" "g8_b64:B=64,L=8" "g12_b32:B=32,L=12" "g4_b128:B=128,L=4" \
  "attn_only_g1:B=128,L=1,av=2" "attn_only_g3:B=128,L=3,av=2" "attn_only_g8b48:B=48,L=8,av=2" \
  "gemm_only_g1:B=128,L=1,av=1" "gemm_only_g3:B=128,L=3,av=1" "gemm_only_g8b48:B=48,L=8,av=1" \
  > $O/steps.jsonl 2> $O/steps.err; rc=$?
python - <<'PY'
import json
for l in open('gpurun_out/r2c5/steps.jsonl'):
    d=json.loads(l); print(f"{d['config']:18s} B={d['B']:3d} L={d['lanes']:2d} skip={d['dbg_skip']} ms/step min={min(d['ms_per_step_lane']):.3f} max={max(d['ms_per_step_lane']):.3f} tok/s={d['tokens_per_s_device_time']} {d['errors'][:1]}")
PY
tail -3 $O/steps.err; echo "microbench rc=$rc"
[ $rc -ge 124 ] && exit 1
for cfg in "3 128" "4 96" "6 64" "8 48"; do
  set -- $cfg
  timeout -k 10 400 python bench.py --lanes $1 --lane-streams $2 --sample-tokens 4096 --steps 4 --warmup 1 --cpu-sample 0 --profile-tokens 0 > $O/bench_l$1.json 2> $O/bench_l$1.err; rc=$?
  python - $O/bench_l$1.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print(d['config']['decode_lanes'], d['config']['streams_per_lane'], 'value', d['value'], 'compress', d['compress_tokens_per_s_rank0'], 'decompress', d['decompress_tokens_per_s_rank0'])
PY
  [ $rc -ge 124 ] && exit 1
done
exit 0
