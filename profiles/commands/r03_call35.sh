#!/bin/bash
# round-3 GPU call 35: final evidence at HEAD: the whole GPU suite + smoke, then the rocprofv3 kernel trace and the two PMC passes of the SmolLM profile pass (final kernels)
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3c35; mkdir -p $O
cd $R
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?
tail -4 $O/pytest.log; echo "pytest rc=$rc"
[ $rc -ne 0 ] && exit 1
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; rc=$?; tail -2 $O/smoke.log; [ $rc -ne 0 ] && exit 1
cd /tmp && export TMPDIR=/tmp
run() { local name=$1; shift; timeout -k 10 900 rocprofv3 "$@" > $O/$name.out 2> $O/$name.err; local rc=$?; echo "$name rc=$rc"; grep "profile_pass\]" $O/$name.err | tail -1; return $rc; }
run kt_smollm --kernel-trace --output-format csv -d $O/kt_smollm -- python3 $R/tools/profile_pass.py smollm || exit 1
python3 $R/tools/prof_summary.py $O/kt_smollm $O/r03_pass2048_s512 --skip-frac 0.25; rm -rf $O/kt_smollm; head -12 $O/r03_pass2048_s512_kernel_stats.csv
run pmc_f_smollm --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_f_smollm -- python3 $R/tools/profile_pass.py smollm --tokens 1024 || exit 1
python3 $R/tools/prof_summary.py $O/pmc_f_smollm $O/r03_pmc1024_FETCH_SIZE --skip-frac 0.5; rm -rf $O/pmc_f_smollm
run pmc_w_smollm --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_w_smollm -- python3 $R/tools/profile_pass.py smollm --tokens 1024 || exit 1
python3 $R/tools/prof_summary.py $O/pmc_w_smollm $O/r03_pmc1024_WRITE_SIZE --skip-frac 0.5; rm -rf $O/pmc_w_smollm
ls $O
