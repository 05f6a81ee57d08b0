#!/bin/bash
# round-4 GPU call 12: where the 74 us of the single-row DECODE CDF kernel go (make timers; one stream, 100 steps)
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4c12; mkdir -p $O
cd $R
CANDLEZIP_AMD_LIB=$R/candlezip_amd/lib/libcandlezip_hip_timers.so timeout -k 10 100 python3 tools/cdf_phases.py 49152 0 100 decode > $O/cdf_decode_phases.txt 2> $O/err.txt || { tail -3 $O/err.txt; exit 1; }
CANDLEZIP_AMD_LIB=$R/candlezip_amd/lib/libcandlezip_hip_timers.so timeout -k 10 100 python3 tools/cdf_phases.py 65536 1 100 decode >> $O/cdf_decode_phases.txt 2>> $O/err.txt || { tail -3 $O/err.txt; exit 1; }
cat $O/cdf_decode_phases.txt
