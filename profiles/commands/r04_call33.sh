#!/bin/bash
# round-4 GPU call 33: big-M GEMM blocks started a hashed 0..3 x 127 x 64 cycles (stagger 1: up to ~24 k cycles = 3 K tiles ... ) apart, so that the blocks of a CU do not hit their barriers together
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4c33; mkdir -p $O
cd $R
for lib in libcandlezip_hip.so libcandlezip_hip_gemmstag1.so libcandlezip_hip_gemmstag4.so libcandlezip_hip.so libcandlezip_hip_gemmstag1.so; do
  CANDLEZIP_AMD_LIB=$R/candlezip_amd/lib/$lib timeout -k 10 300 python3 tools/encode_only.py 192 2048 2 > $O/enc.json 2>> $O/err.txt || { tail -3 $O/err.txt; exit 1; }
  python3 -c "
import json; d=json.load(open('$O/enc.json')); print('$lib', d['compress_tokens_per_s'], d['families_tflops'])"
done
