#!/bin/bash
# round-2 GPU call 13: the real pipeline for the wide-step forms at 2 x 512 streams
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2c13; mkdir -p $O
cd $R
run() { # tag lanes lane_streams streams extra
  timeout -k 10 500 python bench.py --lanes $2 --lane-streams $3 --streams $4 --window 512 --sample-tokens 4096 --steps 6 --warmup 1 --cpu-sample 0 --profile-tokens 0 $5 > $O/bench_$1.json 2> $O/bench_$1.err; rc=$?
  python - $O/bench_$1.json "$1" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print(sys.argv[2], 'streams', d['config']['streams_per_gpu'], 'value', d['value'], 'compress', d['compress_tokens_per_s_rank0'], 'decompress', d['decompress_tokens_per_s_rank0'], 'ms/step', d['ms_per_step'])
except Exception as e: print(sys.argv[2], 'FAILED', e)
PY
  [ $rc -ge 124 ] && exit 1
  return 0
}
run tiled_2x512 2 512 1024 "--debug-set wide_mode=1"
run hybrid_2x512 2 512 1024 "--debug-set wide_mode=2"
run mid_2x512 2 512 1024 "--debug-set wide_mode=3"
run hybrid_2x384 2 384 768 "--debug-set wide_mode=2"
run hybrid_3x341 3 342 1024 "--debug-set wide_mode=2"
exit 0
