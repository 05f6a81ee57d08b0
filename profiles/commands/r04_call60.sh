#!/bin/bash
# round-4 GPU call 60: inside a score chunk of the prefill attention (tuning build): wait for the K registers / staging + barrier / reads + MFMAs + masks
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4c60; mkdir -p $O
cd $R
CANDLEZIP_AMD_LIB=$R/candlezip_amd/lib/libcandlezip_hip_attndbg.so CZ_ATTN_DBG=0 timeout -k 10 300 python3 tools/encode_only.py 192 1024 2 > $O/enc_dbg.json 2>> $O/err.txt || { tail -3 $O/err.txt; exit 1; }
tail -1 $O/enc_dbg.json
