#!/bin/bash
# round-3 GPU call 40: cdf_par_kernel with the row streamed coalesced through LDS (was: every thread loads its own consecutive logits): CDF parity tests, kernel durations new vs previous library
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3c40; mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -m gpu -x -q -k "cdf or decode_symbol or rwkv or payload or encode or literal or division" > $O/pytest.log 2>&1; rc=$?
tail -3 $O/pytest.log; echo "pytest rc=$rc"
[ $rc -ne 0 ] && exit 1
cd /tmp && export TMPDIR=/tmp
for which in new prev; do
  if [ $which = prev ]; then export CANDLEZIP_AMD_LIB=$R/candlezip_amd/lib/libcandlezip_hip_prev.so; fi
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/kt_$which -- python3 $R/tools/cdf_kernel_times.py 2048 > $O/kt_$which.out 2> $O/kt_$which.err; rc=$?; echo "$which rc=$rc"; tail -1 $O/kt_$which.out
  [ $rc -ne 0 ] && { tail -5 $O/kt_$which.err; exit 1; }
  python3 $R/tools/prof_summary.py $O/kt_$which $O/cdf_$which --skip-frac 0.0001 > /dev/null; rm -rf $O/kt_$which; grep "cdf_par" $O/cdf_${which}_kernel_stats.csv
done
