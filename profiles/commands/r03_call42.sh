#!/bin/bash
# round-3 GPU call 42: literals-mode CDF with the row staged through LDS: the rwkv / cdf tests, the RWKV7 kernel trace, the rwkv7 bench line and the one-stream numbers
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3c42; mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py tests/test_gpu_round2.py tests/test_gpu_round3.py tests/test_gpu_fuzz.py -m gpu -x -q -k "cdf or decode_symbol or rwkv or literal or canterbury or xe or cross" > $O/pytest.log 2>&1; rc=$?
tail -3 $O/pytest.log; echo "pytest rc=$rc"
[ $rc -ne 0 ] && exit 1
timeout -k 10 300 python tools/bench_rwkv7.py 1 2048 1 1 > $O/rwkv7_b1.jsonl 2> $O/rwkv7_b1.err; cut -c1-260 $O/rwkv7_b1.jsonl
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --output-format csv -d $O/kt -- python3 $R/tools/profile_pass.py rwkv7 > $O/kt.out 2> $O/kt.err; rc=$?; echo "kt rc=$rc"; [ $rc -ne 0 ] && exit 1
python3 $R/tools/prof_summary.py $O/kt $O/r03_passrwkv7_s1024_t256 --skip-frac 0.0001 > /dev/null; rm -rf $O/kt; head -10 $O/r03_passrwkv7_s1024_t256_kernel_stats.csv
cd $R
timeout -k 10 600 python bench.py --backend rwkv7 --steps 2 --warmup 1 > $O/bench_rwkv7.json 2> $O/bench_rwkv7.err; rc=$?
cut -c1-200 $O/bench_rwkv7.json; echo "rwkv7 rc=$rc"
exit $rc
