#!/bin/bash
# round-4 GPU call 57: rocprofv3 kernel trace of the SmolLM profile pass (tools/profile_pass.py: eager, one host wait per token) at the kernels after the diet
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4c57; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
run() { local name=$1; shift; timeout -k 10 900 rocprofv3 "$@" > $O/$name.out 2> $O/$name.err; local rc=$?; echo "$name rc=$rc"; grep "profile_pass\]" $O/$name.err | tail -1; return $rc; }
run kt_smollm --kernel-trace --output-format csv -d $O/kt_smollm -- python3 $R/tools/profile_pass.py smollm || exit 1
python3 $R/tools/prof_summary.py $O/kt_smollm $O/r04_pass2048_s512_after_diet --skip-frac 0.25; rm -rf $O/kt_smollm; head -24 $O/r04_pass2048_s512_after_diet_kernel_stats.csv
