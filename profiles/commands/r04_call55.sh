#!/bin/bash
# round-4 GPU call 55: sections of a K tile of the big-M GEMM (lm_head tiles; tuning build gemm_clock) at two blocks / one block per CU
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4c55; mkdir -p $O
cd $R
export CANDLEZIP_AMD_LIB=$R/candlezip_amd/lib/libcandlezip_hip_gemmclock.so
for kb in 0; do
  timeout -k 10 300 python3 tools/gemm_clock.py 192 2048 $kb >> $O/gemm_clock.jsonl 2>> $O/err.txt || { tail -3 $O/err.txt; exit 1; }
done
python3 -c "
import json
for l in open('$O/gemm_clock.jsonl'):
    d=json.loads(l); print(d['gemm_lds_kb'], d['gemm_tflops_hip_events'], d['tiles_128x128_lm_head'], d['lm_head_k_tile_sections_cycles'])"
