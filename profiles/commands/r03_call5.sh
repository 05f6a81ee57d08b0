#!/bin/bash
# round-3 GPU call 5: rocprofv3 evidence for the passes bench.py measures its rooflines on (smollm steady-state pass, rwkv7 pass): --kernel-trace,
# then FETCH_SIZE and WRITE_SIZE in passes of their own; then the two bench lines
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3c5; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
run() {  # name, rocprof args..., -- program args
  local name=$1; shift
  timeout -k 10 900 rocprofv3 "$@" > $O/$name.out 2> $O/$name.err; local rc=$?
  echo "$name rc=$rc"; grep "profile_pass\]" $O/$name.err | tail -1
  return $rc
}
run kt_smollm --kernel-trace --output-format csv -d $O/kt_smollm -- python3 $R/tools/profile_pass.py smollm || exit 1
python3 $R/tools/prof_summary.py $O/kt_smollm $O/r03_pass2048_s512 --skip-frac 0.25; rm -rf $O/kt_smollm; head -12 $O/r03_pass2048_s512_kernel_stats.csv
run pmc_f_smollm --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_f_smollm -- python3 $R/tools/profile_pass.py smollm --tokens 1024 || exit 1
python3 $R/tools/prof_summary.py $O/pmc_f_smollm $O/r03_pmc1024_FETCH_SIZE --skip-frac 0.5; rm -rf $O/pmc_f_smollm
run pmc_w_smollm --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_w_smollm -- python3 $R/tools/profile_pass.py smollm --tokens 1024 || exit 1
python3 $R/tools/prof_summary.py $O/pmc_w_smollm $O/r03_pmc1024_WRITE_SIZE --skip-frac 0.5; rm -rf $O/pmc_w_smollm
run kt_rwkv7 --kernel-trace --output-format csv -d $O/kt_rwkv7 -- python3 $R/tools/profile_pass.py rwkv7 || exit 1
python3 $R/tools/prof_summary.py $O/kt_rwkv7 $O/r03_passrwkv7_s256_t512 --skip-frac 0.0001; rm -rf $O/kt_rwkv7; head -14 $O/r03_passrwkv7_s256_t512_kernel_stats.csv
run pmc_f_rwkv7 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_f_rwkv7 -- python3 $R/tools/profile_pass.py rwkv7 || exit 1
python3 $R/tools/prof_summary.py $O/pmc_f_rwkv7 $O/r03_pmcrwkv7_FETCH_SIZE; rm -rf $O/pmc_f_rwkv7
run pmc_w_rwkv7 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_w_rwkv7 -- python3 $R/tools/profile_pass.py rwkv7 || exit 1
python3 $R/tools/prof_summary.py $O/pmc_w_rwkv7 $O/r03_pmcrwkv7_WRITE_SIZE; rm -rf $O/pmc_w_rwkv7
ls -la $O/*.json $O/*.csv
cd $R
timeout -k 10 600 python bench.py --backend rwkv7 --steps 2 --warmup 1 > $O/bench_rwkv7.json 2> $O/bench_rwkv7.err; rc=$?
tail -2 $O/bench_rwkv7.err; cut -c1-300 $O/bench_rwkv7.json; echo "bench rwkv7 rc=$rc"
exit 0
