#!/bin/bash
# round-4 GPU call 67: what a read-only stream reaches in dec_attn's access shapes (tools/microbench/hbm_read.hip)
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4c67; mkdir -p $O
cd $R
timeout -k 10 200 candlezip_amd/lib/hbm_read > $O/hbm_read.jsonl 2> $O/err.txt || { tail -3 $O/err.txt; exit 1; }
cat $O/hbm_read.jsonl
