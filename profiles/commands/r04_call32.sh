#!/bin/bash
# round-4 GPU call 32: the clock the chip holds inside the big-M GEMM's K loop (make gemm_clock: s_memtime / s_memrealtime), i.e. the f32 MFMA peak the kernel can be priced against
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4c32; mkdir -p $O
cd $R
CANDLEZIP_AMD_LIB=$R/candlezip_amd/lib/libcandlezip_hip_gemmclock.so timeout -k 10 300 python3 tools/gemm_clock.py 192 2048 > $O/gemm_clock.json 2> $O/err.txt || { tail -3 $O/err.txt; exit 1; }
cat $O/gemm_clock.json
