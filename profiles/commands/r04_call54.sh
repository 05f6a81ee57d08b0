#!/bin/bash
# round-4 GPU call 54: PIPE for the 128 x 64 tiles too (two blocks per CU with half-tile staging between the MFMAs, against three one-buffer blocks): gemm_pipe 1 / 3
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4c54; mkdir -p $O
cd $R
for rep in 1 2; do for v in 1 3; do
  CZ_BENCH_DEBUG_SET=gemm_pipe=$v timeout -k 10 300 python3 tools/encode_only.py 192 2048 2 > $O/enc.json 2>> $O/err.txt || { tail -3 $O/err.txt; exit 1; }
  python3 -c "
import json; d=json.load(open('$O/enc.json')); print('gemm_pipe $v', d['compress_tokens_per_s'], d['families_tflops'])" | tee -a $O/pipe.txt
done; done
