#!/bin/bash
# round-4 GPU call 11: per-kernel durations of the single-stream step on the VALU projections (rocprofv3 --kernel-trace of the EAGER step with one host
# wait per step: the only profiled form that is safe on this stack), and the CDF kernel's phases for ONE row (make timers)
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4c11; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/kt -- python3 $R/tools/bench_decode_step.py --iters 40 --max-streams 8 b1_trace:B=1,g=0,ss=1 > $O/trace.out 2> $O/trace.err; rc=$?
echo "trace rc=$rc"; tail -2 $O/trace.err | cut -c1-200
[ $rc -ne 0 ] && exit 1
python3 $R/tools/trace_step_kernels.py $O/kt > $O/r04_b1_step_kernel_durations.txt; rm -rf $O/kt; cat $O/r04_b1_step_kernel_durations.txt
cd $R
for rows in 1 2 64; do CANDLEZIP_AMD_LIB=$R/candlezip_amd/lib/libcandlezip_hip_timers.so timeout -k 10 100 python3 tools/cdf_phases.py 49152 0 $rows >> $O/cdf_phases_rows.txt 2>> $O/err.txt || exit 1; done
cat $O/cdf_phases_rows.txt
