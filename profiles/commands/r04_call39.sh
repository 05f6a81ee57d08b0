#!/bin/bash
# round-4 GPU call 39: s_setprio 3 in the big-M GEMM's epilogue (prio1) / epilogue + prologue (prio3): the VALU-only phases at full issue rate at the cost of the partner block's MFMA issue
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4c39; mkdir -p $O
cd $R
for lib in libcandlezip_hip.so libcandlezip_hip_gemmprio1.so libcandlezip_hip_gemmprio3.so libcandlezip_hip.so libcandlezip_hip_gemmprio1.so libcandlezip_hip_gemmprio3.so; do
  CANDLEZIP_AMD_LIB=$R/candlezip_amd/lib/$lib timeout -k 10 300 python3 tools/encode_only.py 192 2048 2 > $O/enc.json 2>> $O/err.txt || { tail -3 $O/err.txt; exit 1; }
  python3 -c "
import json; d=json.load(open('$O/enc.json')); print('$lib', d['compress_tokens_per_s'], d['families_tflops'])" | tee -a $O/prio.txt
done
