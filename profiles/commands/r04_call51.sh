#!/bin/bash
# round-4 GPU call 51: prefill attention diet, final form: full GPU suite, encode at 512 / 1024 / 2048 tokens per stream (<3,8>, <3,16>, <3,16> x 2 windows)
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4c51; mkdir -p $O
cd $R
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1 || { tail -30 $O/pytest_gpu.log; exit 1; }
tail -2 $O/pytest_gpu.log
for T in 512 1024 2048; do timeout -k 10 300 python3 tools/encode_only.py 192 $T 2 > $O/enc$T.json 2>> $O/err.txt || { tail -3 $O/err.txt; exit 1; }; python3 -c "
import json; d=json.load(open('$O/enc$T.json')); print($T, d['compress_tokens_per_s'], d['families_tflops'], d['families_ms']['attention_f32_mfma'])"; done
