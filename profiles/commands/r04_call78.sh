#!/bin/bash
# round-4 GPU call 78: the row maxima of the prefill attention reduced over the 16 lanes of a DPP row on the VALU (quad permutes + mirrors) instead of four ds_bpermute round trips: parity, encode
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4c78; mkdir -p $O
cd $R
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q -k "attention or encode or prefill or kernels" > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -1 $O/pytest.log
for T in 1024 2048; do timeout -k 10 300 python3 tools/encode_only.py 192 $T 2 > $O/enc$T.json 2>> $O/err.txt || { tail -3 $O/err.txt; exit 1; }; python3 -c "
import json; d=json.load(open('$O/enc$T.json')); print($T, d['compress_tokens_per_s'], d['families_tflops'], d['families_ms']['attention_f32_mfma'])"; done
