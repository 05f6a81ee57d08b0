#!/bin/bash
# round-3 GPU call 45: rw_state_kernel with non-temporal loads / stores of its state tile: rwkv parity, kernel durations new vs previous library (profile pass under rocprofv3)
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3c45; mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_model.py tests/test_gpu_round3.py -m gpu -x -q -k "rwkv" > $O/pytest.log 2>&1; rc=$?
tail -2 $O/pytest.log; echo "pytest rc=$rc"; [ $rc -ne 0 ] && exit 1
cd /tmp && export TMPDIR=/tmp
for which in new prev new prev; do
  if [ $which = prev ]; then export CANDLEZIP_AMD_LIB=$R/candlezip_amd/lib/libcandlezip_hip_prev.so; else unset CANDLEZIP_AMD_LIB; fi
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/kt -- python3 $R/tools/profile_pass.py rwkv7 > $O/kt.out 2> $O/kt.err; rc=$?; [ $rc -ne 0 ] && { tail -3 $O/kt.err; exit 1; }
  python3 $R/tools/prof_summary.py $O/kt $O/pass_$which --skip-frac 0.0001 > /dev/null; rm -rf $O/kt; echo "$which: $(grep rw_state_kernel $O/pass_${which}_kernel_stats.csv)"
done
