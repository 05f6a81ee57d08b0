#!/bin/bash
# round-4 GPU call 73: lane shapes of the self-test line once more at the faster prefill kernels: 2 x 512 (default), 4 x 256, 3 x 352 streams (K = 6)
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4c73; mkdir -p $O
cd $R
run() { timeout -k 10 400 python3 bench.py --streams $1 --lanes $2 --lane-streams $3 --steps 6 --warmup 2 --cpu-sample 0 --profile-tokens 0 --single-stream-tokens 0 > $O/bench_s$1_l$2.json 2> $O/bench_s$1_l$2.err; rc=$?; python3 -c "
import json; d=json.load(open('$O/bench_s$1_l$2.json')); print('$1 $2 $3', d['value'], d['compress_tokens_per_s_rank0'])" || tail -3 $O/bench_s$1_l$2.err; return $rc; }
run 1024 4 256 || exit 1
run 1056 3 352 || exit 1

