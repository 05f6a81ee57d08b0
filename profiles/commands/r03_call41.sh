#!/bin/bash
# round-3 GPU call 41: kernel durations of the CDF kernels, new (row staged through LDS) vs previous library, 2048 and 8192 rows
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3c41; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for rows in 2048 8192; do
for which in new prev; do
  if [ $which = prev ]; then export CANDLEZIP_AMD_LIB=$R/candlezip_amd/lib/libcandlezip_hip_prev.so; else unset CANDLEZIP_AMD_LIB; fi
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/kt_$which -- python3 $R/tools/cdf_kernel_times.py $rows > $O/kt_$which.out 2> $O/kt_$which.err; rc=$?; echo "$which rows=$rows rc=$rc"
  [ $rc -ne 0 ] && { tail -5 $O/kt_$which.err; exit 1; }
  python3 $R/tools/prof_summary.py $O/kt_$which $O/cdf_${which}_$rows --skip-frac 0.0001 > /dev/null; rm -rf $O/kt_$which; grep "cdf_par" $O/cdf_${which}_${rows}_kernel_stats.csv
done
done
