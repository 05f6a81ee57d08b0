#!/bin/bash
# round-4 GPU call 40: what the f32 matrix pipe delivers with nothing else in the way (tools/microbench/mfma_peak.hip)
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4c40; mkdir -p $O
cd $R
timeout -k 10 200 candlezip_amd/lib/mfma_peak > $O/mfma_peak.jsonl 2> $O/err.txt || { tail -3 $O/err.txt; exit 1; }
cat $O/mfma_peak.jsonl
