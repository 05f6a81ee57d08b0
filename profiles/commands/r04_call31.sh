#!/bin/bash
# round-4 GPU call 31: the big-M GEMM : the 128 x 64 tiles compiled for three blocks per CU (make gemm_n64occ3: 138 VGPRs, 52 KB of LDS) against two, encode only
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4c31; mkdir -p $O
cd $R
for lib in libcandlezip_hip.so libcandlezip_hip_n64occ3.so libcandlezip_hip.so libcandlezip_hip_n64occ3.so; do
  CANDLEZIP_AMD_LIB=$R/candlezip_amd/lib/$lib timeout -k 10 300 python3 tools/encode_only.py 192 2048 2 > $O/enc.json 2>> $O/err.txt || { tail -3 $O/err.txt; exit 1; }
  python3 -c "
import json; d=json.load(open('$O/enc.json')); print('$lib', d['compress_tokens_per_s'], d['families_tflops'])"
done
