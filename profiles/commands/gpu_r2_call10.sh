#!/bin/bash
# round-2 GPU call 10: the real pipeline (re-primes included) for wide lock-step groups
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2c10; mkdir -p $O
cd $R
run() { # lanes lane_streams streams window extra
  tag="l$1x$2$5"; tag=${tag// /_}; tag=${tag//=/}
  timeout -k 10 500 python bench.py --lanes $1 --lane-streams $2 --streams $3 --window $4 --sample-tokens 4096 --steps 6 --warmup 1 --cpu-sample 0 --profile-tokens 0 $5 > $O/bench_$tag.json 2> $O/bench_$tag.err; rc=$?
  python - $O/bench_$tag.json "$tag" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print(sys.argv[2], 'streams', d['config']['streams_per_gpu'], 'value', d['value'], 'compress', d['compress_tokens_per_s_rank0'], 'decompress', d['decompress_tokens_per_s_rank0'], 'ms/step', d['ms_per_step'])
except Exception as e: print(sys.argv[2], 'FAILED', e)
PY
  [ $rc -ge 124 ] && exit 1
  return 0
}
run 2 384 768 512 ""
run 2 384 768 512 "--debug-set wide_min_streams=0"
run 2 512 1024 512 ""
run 1 768 768 512 ""
run 1 1024 1024 512 ""
exit 0
