#!/bin/bash
# round-3 GPU call 23: CDF kernels with div_by() (reciprocal + fma corrections instead of the f64 division): parity, then the kernel traces of both passes
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3c23; mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py tests/test_gpu_fuzz.py -m gpu -x -q -k "division or cdf or decode_symbol or rwkv or xe or literal or fuzz or payload" > $O/pytest.log 2>&1; rc=$?
tail -4 $O/pytest.log; echo "pytest rc=$rc"
[ $rc -ne 0 ] && exit 1
cd /tmp && export TMPDIR=/tmp
run() { local name=$1; shift; timeout -k 10 900 rocprofv3 "$@" > $O/$name.out 2> $O/$name.err; local rc=$?; echo "$name rc=$rc"; grep "profile_pass\]" $O/$name.err | tail -1; return $rc; }
run kt_rwkv7 --kernel-trace --output-format csv -d $O/kt_rwkv7 -- python3 $R/tools/profile_pass.py rwkv7 || exit 1
python3 $R/tools/prof_summary.py $O/kt_rwkv7 $O/r03_passrwkv7_s256_t512 --skip-frac 0.0001; rm -rf $O/kt_rwkv7; head -12 $O/r03_passrwkv7_s256_t512_kernel_stats.csv
run kt_smollm --kernel-trace --output-format csv -d $O/kt_smollm -- python3 $R/tools/profile_pass.py smollm || exit 1
python3 $R/tools/prof_summary.py $O/kt_smollm $O/r03_pass2048_s512 --skip-frac 0.25; rm -rf $O/kt_smollm; grep "cdf\|xe_" $O/r03_pass2048_s512_kernel_stats.csv
exit 0
