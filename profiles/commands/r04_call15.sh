#!/bin/bash
# round-4 GPU call 15: gemv_max_rows = 16 as shipped: the whole GPU suite; RWKV7 8 / 16 streams with and without the VALU projections
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4c15; mkdir -p $O
cd $R && timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?
tail -3 $O/pytest.log
[ $rc -ne 0 ] && exit 1
for B in 8 16 32; do
  for gv in 0 64; do
    CZ_BENCH_DEBUG_SET=gemv_max_rows=$gv timeout -k 10 200 python3 tools/bench_rwkv7.py $B 1024 >> $O/rwkv7_groups.jsonl 2>> $O/err.txt || { tail -3 $O/err.txt; exit 1; }
  done
done
cut -c1-330 $O/rwkv7_groups.jsonl
