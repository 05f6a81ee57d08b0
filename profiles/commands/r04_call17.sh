#!/bin/bash
# round-4 GPU call 17: prefill attention: the three-pass kernel (v3: nothing kept, QK^T three times) everywhere against the register-resident v2
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4c17; mkdir -p $O
cd $R
for v in 0 2 0 2; do
  timeout -k 10 300 python3 tools/encode_only.py 192 2048 2 $v > $O/enc_v$v.json 2>> $O/err.txt || { tail -3 $O/err.txt; exit 1; }
  python3 -c "
import json; d=json.load(open('$O/enc_v$v.json')); print('attn_v1=$v', d['compress_tokens_per_s'], d['families_ms']['attention_f32_mfma'], d['attention_tflops'])"
done
