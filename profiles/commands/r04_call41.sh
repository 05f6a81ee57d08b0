#!/bin/bash
# round-4 GPU call 41: what the exponentials / divisions / masks of the prefill attention cost (tuning build attn_dbg: phases switched off, results wrong), at the bench's shape (512-token windows after 512 of history: <3,16>) and at 2048 tokens (<3,32>)
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4c41; mkdir -p $O
cd $R
export CANDLEZIP_AMD_LIB=$R/candlezip_amd/lib/libcandlezip_hip_attndbg.so
for T in 1024 2048; do for m in 0 2 8 10 1 16 27; do
  CZ_ATTN_DBG=$m timeout -k 10 300 python3 tools/encode_only.py 192 $T 2 > $O/enc.json 2>> $O/err.txt || { tail -3 $O/err.txt; exit 1; }
  python3 -c "
import json; d=json.load(open('$O/enc.json')); print('tokens $T mask $m', d['compress_tokens_per_s'], 'attention ms', d['families_ms']['attention_f32_mfma'], 'TF', d['families_tflops']['attention_f32_mfma'])" | tee -a $O/attn_dbg.txt
done; done
