#!/bin/bash
# round-4 GPU call 35: cycles of the big-M GEMM's phases (tuning build gemm_clock): K loop per K tile at two blocks / one block per CU, prologue, epilogue
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4c35; mkdir -p $O
cd $R
export CANDLEZIP_AMD_LIB=$R/candlezip_amd/lib/libcandlezip_hip_gemmclock.so
for kb in 0 100; do
  timeout -k 10 300 python3 tools/gemm_clock.py 192 2048 $kb >> $O/gemm_clock.jsonl 2>> $O/err.txt || { tail -3 $O/err.txt; exit 1; }
done
cat $O/gemm_clock.jsonl
