#!/bin/bash
# round-3 GPU call 13: phase stamps inside gemm_mid_kernel (wide decode step, 512 and 1024 streams)
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3c13; mkdir -p $O
cd $R
export CANDLEZIP_AMD_LIB=$R/candlezip_amd/lib/libcandlezip_hip_middbg.so
timeout -k 10 300 python tools/mid_phases.py 512 767 20 > $O/mid_phases_512.json 2> $O/mid_phases_512.err; cat $O/mid_phases_512.json; tail -2 $O/mid_phases_512.err
timeout -k 10 300 python tools/mid_phases.py 1024 767 10 > $O/mid_phases_1024.json 2> $O/mid_phases_1024.err; cat $O/mid_phases_1024.json | head -70
exit 0
