#!/bin/bash
# round-2 GPU call 20: compress beside decompress (two sessions sharing the device) against one after the other
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2c20; mkdir -p $O
cd $R
for ov in 1 0; do
  timeout -k 10 500 python bench.py --overlap $ov --sample-tokens 4096 --steps 8 --warmup 1 --cpu-sample 0 --profile-tokens 0 > $O/bench_ov$ov.json 2> $O/bench_ov$ov.err; rc=$?
  python - $O/bench_ov$ov.json $ov <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print('overlap', sys.argv[2], 'value', d['value'], 'compress', d['compress_tokens_per_s_rank0'], 'decompress', d['decompress_tokens_per_s_rank0'], 'ms/step', d['ms_per_step'])
except Exception as e: print('FAILED', e)
PY
  tail -2 $O/bench_ov$ov.err | cut -c1-300
  [ $rc -ge 124 ] && exit 1
done
exit 0
