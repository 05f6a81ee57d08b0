#!/bin/bash
# round-3 GPU call 47: the rwkv7 line at the final head (staged literals-mode CDF, non-temporal state tile)
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3c47; mkdir -p $O
cd $R
timeout -k 10 600 python bench.py --backend rwkv7 --steps 2 --warmup 1 > $O/bench_rwkv7.json 2> $O/bench_rwkv7.err; rc=$?
cut -c1-200 $O/bench_rwkv7.json; echo "rwkv7 rc=$rc"
exit $rc
