#!/bin/bash
# round-4 GPU call 64: cz_expf2 (two exponentials on packed f32 multiply-adds) in the prefill attention and the SwiGLU epilogue: exhaustive test of both halves, parity tests, encode
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4c64; mkdir -p $O
cd $R
timeout -k 10 300 python3 -m pytest tests/test_gpu_round4.py -m gpu -x -q -k "expf" > $O/pytest_expf.log 2>&1 || { tail -30 $O/pytest_expf.log; exit 1; }
tail -1 $O/pytest_expf.log
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q -k "attention or encode or prefill or segment or batch or parity or kernels or reprime or round_trip" > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
for T in 512 1024 2048; do timeout -k 10 300 python3 tools/encode_only.py 192 $T 2 > $O/enc$T.json 2>> $O/err.txt || { tail -3 $O/err.txt; exit 1; }; python3 -c "
import json; d=json.load(open('$O/enc$T.json')); print($T, d['compress_tokens_per_s'], d['families_tflops'], d['families_ms']['attention_f32_mfma'])"; done
