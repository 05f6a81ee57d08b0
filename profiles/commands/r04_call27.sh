#!/bin/bash
# round-4 GPU call 27: the big-M GEMM with one accumulator chain after the other (the chunk partial of tile (i, j) added under the chain of the next tile: make gemm_accmajor) against the shipped order, encode only
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4c27; mkdir -p $O
cd $R
for lib in libcandlezip_hip.so libcandlezip_hip_accmajor.so libcandlezip_hip.so libcandlezip_hip_accmajor.so; do
  CANDLEZIP_AMD_LIB=$R/candlezip_amd/lib/$lib timeout -k 10 300 python3 tools/encode_only.py 192 2048 2 > $O/enc.json 2>> $O/err.txt || { tail -3 $O/err.txt; exit 1; }
  python3 -c "
import json; d=json.load(open('$O/enc.json')); print('$lib', d['compress_tokens_per_s'], d['families_tflops'])"
done
