#!/bin/bash
# round-3 GPU call 50: the two PMC passes of the RWKV7 profile pass at the final kernels (staged literals-mode CDF, non-temporal state tile)
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3c50; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
run() { local name=$1; shift; timeout -k 10 900 rocprofv3 "$@" > $O/$name.out 2> $O/$name.err; local rc=$?; echo "$name rc=$rc"; grep "profile_pass" $O/$name.err | tail -1; return $rc; }
run pmc_f_rwkv7 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_f_rwkv7 -- python3 $R/tools/profile_pass.py rwkv7 || exit 1
python3 $R/tools/prof_summary.py $O/pmc_f_rwkv7 $O/r03_pmcrwkv7_FETCH_SIZE > /dev/null; rm -rf $O/pmc_f_rwkv7
run pmc_w_rwkv7 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_w_rwkv7 -- python3 $R/tools/profile_pass.py rwkv7 || exit 1
python3 $R/tools/prof_summary.py $O/pmc_w_rwkv7 $O/r03_pmcrwkv7_WRITE_SIZE > /dev/null; rm -rf $O/pmc_w_rwkv7
ls $O
