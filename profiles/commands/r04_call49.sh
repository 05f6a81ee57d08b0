#!/bin/bash
# round-4 GPU call 49: cycles of the prefill attention's phases (tuning build attn_dbg with stamps), 192 x 1024 tokens (<3,16>) and 192 x 2048 (<3,32>); with and without the debug masks
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4c49; mkdir -p $O
cd $R
export CANDLEZIP_AMD_LIB=$R/candlezip_amd/lib/libcandlezip_hip_attndbg.so
for T in 1024 2048; do for m in 0 27; do
  CZ_ATTN_DBG=$m timeout -k 10 300 python3 tools/encode_only.py 192 $T 2 > $O/enc_${T}_$m.json 2>> $O/err.txt || { tail -3 $O/err.txt; exit 1; }
  echo "tokens $T mask $m"; tail -1 $O/enc_${T}_$m.json
done; done
