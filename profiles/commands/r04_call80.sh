#!/bin/bash
# round-4 GPU call 80: the prefill attention as two instantiations (row-layout K for the encode, transposed K for the re-prime): parity subset, encode timing
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4c80; mkdir -p $O
cd $R
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q -k "reprime or attention or kernels or round4" > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -1 $O/pytest.log
for T in 1024 2048; do timeout -k 10 300 python3 tools/encode_only.py 192 $T 2 > $O/enc$T.json 2>> $O/err.txt || { tail -3 $O/err.txt; exit 1; }; python3 -c "
import json; d=json.load(open('$O/enc$T.json')); print($T, d['compress_tokens_per_s'], d['families_tflops'], d['families_ms']['attention_f32_mfma'])"; done
