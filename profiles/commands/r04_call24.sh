#!/bin/bash
# round-4 GPU call 24: RWKV7 groups of 96 .. 256 streams: the narrow step (dec_gemm_kernel) against the wide one (grouped gemm_mid launches)
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4c24; mkdir -p $O
cd $R
for B in 96 128 160 192; do
  for wm in 100000 1; do
    CZ_BENCH_DEBUG_SET=rwkv_wide_min=$wm timeout -k 10 200 python3 tools/bench_rwkv7.py $B 512 >> $O/rwkv7_mid_groups.jsonl 2>> $O/err.txt || { tail -3 $O/err.txt; exit 1; }
  done
done
python3 - <<'PY'
import json,os
for l in open(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r4c24/rwkv7_mid_groups.jsonl"):
    d=json.loads(l); print(d["streams"], d["debug_set"], d["decompress_tokens_per_s"], d["compress_tokens_per_s"])
PY
