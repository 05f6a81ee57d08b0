#!/bin/bash
# round-4 GPU call 9: RWKV7 groups of 1..4 streams on the VALU projections: parity (tiny and 0.1B shapes, gemv on / off; the existing RWKV7 tests whose
# single-stream steps now take them), then one / two / four streams x 2048 tokens at the 0.1B shape with and without
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4c9; mkdir -p $O
cd $R && timeout -k 10 800 python3 -m pytest tests/test_gpu_round4.py tests/test_gpu_round3.py tests/test_gpu_model.py tests/test_gpu_fuzz.py -m gpu -x -q -k "rwkv" > $O/pytest.log 2>&1; rc=$?
tail -3 $O/pytest.log
[ $rc -ne 0 ] && exit 1
for B in 1 2 4; do
  for gv in 0 4; do
    CZ_BENCH_DEBUG_SET=gemv_max_rows=$gv timeout -k 10 200 python3 tools/bench_rwkv7.py $B 2048 >> $O/rwkv7_small_groups.jsonl 2>> $O/err.txt || { tail -3 $O/err.txt; exit 1; }
  done
done
cat $O/rwkv7_small_groups.jsonl | cut -c1-400
