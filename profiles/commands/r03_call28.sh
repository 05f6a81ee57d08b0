#!/bin/bash
# round-3 GPU call 28: the two PMC passes of the RWKV7 profile pass at its new shape (1 024 streams x 256 tokens), then the rwkv7 bench line
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3c28; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
run() { local name=$1; shift; timeout -k 10 900 rocprofv3 "$@" > $O/$name.out 2> $O/$name.err; local rc=$?; echo "$name rc=$rc"; grep "profile_pass" $O/$name.err | tail -1; return $rc; }
run pmc_f_rwkv7 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_f_rwkv7 -- python3 $R/tools/profile_pass.py rwkv7 || exit 1
python3 $R/tools/prof_summary.py $O/pmc_f_rwkv7 $O/r03_pmcrwkv7_FETCH_SIZE; rm -rf $O/pmc_f_rwkv7
run pmc_w_rwkv7 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_w_rwkv7 -- python3 $R/tools/profile_pass.py rwkv7 || exit 1
python3 $R/tools/prof_summary.py $O/pmc_w_rwkv7 $O/r03_pmcrwkv7_WRITE_SIZE; rm -rf $O/pmc_w_rwkv7
ls $O
cd $R
cp $O/r03_pmcrwkv7_FETCH_SIZE_pmc.json $O/r03_pmcrwkv7_WRITE_SIZE_pmc.json profiles/
timeout -k 10 600 python bench.py --backend rwkv7 --steps 2 --warmup 1 > $O/bench_rwkv7.json 2> $O/bench_rwkv7.err; rc=$?
tail -2 $O/bench_rwkv7.err; cut -c1-300 $O/bench_rwkv7.json; echo "bench rwkv7 rc=$rc"
exit $rc
