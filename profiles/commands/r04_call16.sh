#!/bin/bash
# round-4 GPU call 16: the big-M GEMM by tile family (128 x 128 for gate|up and the lm_head; 128 x 64 for QKV / o / down) in an encode-only run
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4c16; mkdir -p $O
cd $R
timeout -k 10 300 python3 tools/encode_only.py 192 2048 2 > $O/encode_only.json 2> $O/err.txt || { tail -3 $O/err.txt; exit 1; }
python3 -c "
import json; d=json.load(open('$O/encode_only.json')); print(d['compress_tokens_per_s'], d['gemm_tflops'], d['attention_tflops']); print(d['families_ms']); print(d['families_tflops']); print(d['families_launches'])"
