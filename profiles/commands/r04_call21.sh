#!/bin/bash
# round-4 GPU call 21: more streams per GPU (the KV cache of 2 048 streams is 193 GB of the 288): 4 lanes x 512 and 3 lanes x 512 against the default 2 x 512, K = 6
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4c21; mkdir -p $O
cd $R
for cfg in "1024 2" "2048 4" "1536 3"; do
  set -- $cfg
  timeout -k 10 500 python3 bench.py --streams $1 --lanes $2 --lane-streams 512 --steps 6 --warmup 2 --cpu-sample 0 --profile-tokens 0 --single-stream-tokens 0 > $O/bench_s$1_l$2.json 2> $O/bench_s$1_l$2.err; rc=$?
  tail -1 $O/bench_s$1_l$2.err | cut -c1-200; [ $rc -ne 0 ] && exit 1
  python3 -c "
import json;d=json.loads(open('$O/bench_s$1_l$2.json').read().strip().splitlines()[-1]);print('streams $1 lanes $2:', d['value'], d['compress_tokens_per_s_rank0'], d['decompress_tokens_per_s_rank0'], d['ms_per_step'])"
done
