#!/bin/bash
# round-4 GPU call 71: this round's rocprofv3 evidence at the FINAL kernels and block shapes (tools/profile_pass.py: eager, one host wait per token; four-wave decode-attention blocks as bench.py's lanes run them):
# kernel trace + FETCH_SIZE / WRITE_SIZE passes of the SmolLM pass
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4c71; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
run() { local name=$1; shift; timeout -k 10 900 rocprofv3 "$@" > $O/$name.out 2> $O/$name.err; local rc=$?; echo "$name rc=$rc"; grep "profile_pass\]" $O/$name.err | tail -1; return $rc; }
run kt_smollm --kernel-trace --output-format csv -d $O/kt_smollm -- python3 $R/tools/profile_pass.py smollm || exit 1
python3 $R/tools/prof_summary.py $O/kt_smollm $O/r04_pass2048_s512 --skip-frac 0.25; rm -rf $O/kt_smollm; head -6 $O/r04_pass2048_s512_kernel_stats.csv
run pmc_f_smollm --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_f_smollm -- python3 $R/tools/profile_pass.py smollm --tokens 1024 || exit 1
python3 $R/tools/prof_summary.py $O/pmc_f_smollm $O/r04_pmc1024_FETCH_SIZE --skip-frac 0.5; rm -rf $O/pmc_f_smollm
run pmc_w_smollm --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_w_smollm -- python3 $R/tools/profile_pass.py smollm --tokens 1024 || exit 1
python3 $R/tools/prof_summary.py $O/pmc_w_smollm $O/r04_pmc1024_WRITE_SIZE --skip-frac 0.5; rm -rf $O/pmc_w_smollm
ls $O
