#!/bin/bash
# round-4 GPU call 61: rocprofv3 kernel trace of the RWKV7 profile pass at the kernels after the diet
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4c61; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
run() { local name=$1; shift; timeout -k 10 900 rocprofv3 "$@" > $O/$name.out 2> $O/$name.err; local rc=$?; echo "$name rc=$rc"; grep "profile_pass\]" $O/$name.err | tail -1; return $rc; }
run kt_rwkv7 --kernel-trace --output-format csv -d $O/kt_rwkv7 -- python3 $R/tools/profile_pass.py rwkv7 || exit 1
python3 $R/tools/prof_summary.py $O/kt_rwkv7 $O/r04_passrwkv7_s1024_t256_after_diet --skip-frac 0.0001; rm -rf $O/kt_rwkv7; head -20 $O/r04_passrwkv7_s1024_t256_after_diet_kernel_stats.csv
