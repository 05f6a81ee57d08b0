#!/bin/bash
# round-4 GPU call 50: prefill attention v2 with buffer-addressed K/V chunk loads (one v_min clamp, no zero-fill), the causal mask against per-row limits, no row-validity selects: parity, phases, encode
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4c50; mkdir -p $O
cd $R
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q -k "attention or encode or prefill or segment or batch or parity or kernels or reprime or round_trip" > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
CANDLEZIP_AMD_LIB=$R/candlezip_amd/lib/libcandlezip_hip_attndbg.so CZ_ATTN_DBG=0 timeout -k 10 300 python3 tools/encode_only.py 192 1024 2 > $O/enc_dbg.json 2>> $O/err.txt || { tail -3 $O/err.txt; exit 1; }
tail -1 $O/enc_dbg.json
for i in 1 2; do timeout -k 10 300 python3 tools/encode_only.py 192 2048 2 > $O/enc$i.json 2>> $O/err.txt || { tail -3 $O/err.txt; exit 1; }; python3 -c "
import json; d=json.load(open('$O/enc$i.json')); print(d['compress_tokens_per_s'], d['families_tflops'], d['families_ms']['attention_f32_mfma'])"; done
