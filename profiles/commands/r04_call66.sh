#!/bin/bash
# round-4 GPU call 66: phases of gemm_mid_kernel in the 512-stream wide step (tuning build mid_dbg), at this round's addressing
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4c66; mkdir -p $O
cd $R
CANDLEZIP_AMD_LIB=$R/candlezip_amd/lib/libcandlezip_hip_middbg.so timeout -k 10 300 python3 tools/mid_phases.py 512 767 20 > $O/mid_phases.json 2> $O/err.txt || { tail -3 $O/err.txt; exit 1; }
cat $O/mid_phases.json
