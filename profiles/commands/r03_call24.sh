#!/bin/bash
# round-3 GPU call 24: phase timers inside cdf_par_kernel (tuning build), both modes
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3c24; mkdir -p $O
cd $R
export CANDLEZIP_AMD_LIB=$R/candlezip_amd/lib/libcandlezip_hip_timers.so
for cfg in "49152 0 512" "65536 1 512" "65536 1 256" "49152 0 8192"; do timeout -k 10 200 python tools/cdf_phases.py $cfg >> $O/phases.txt 2>> $O/phases.err || exit 1; done
cat $O/phases.txt
