#!/bin/bash
# round-3 GPU call 30: RWKV7-0.1B shape, ONE stream (BASELINE config 4 as the reference CLI runs a file) and small groups: tokens/s; kernel durations of the B = 1 step
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3c30; mkdir -p $O
cd $R
for shape in "1 2048 1 1" "4 2048 4 1" "16 2048 16 1" "64 2048 64 1"; do
  timeout -k 10 300 python tools/bench_rwkv7.py $shape >> $O/rwkv7_small.jsonl 2>> $O/rwkv7_small.err || { echo "shape $shape failed"; tail -3 $O/rwkv7_small.err; exit 1; }
done
cut -c1-300 $O/rwkv7_small.jsonl
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/kt -- python3 $R/tools/bench_rwkv7.py 1 512 1 1 > $O/kt.out 2> $O/kt.err; rc=$?; echo "kt rc=$rc"
[ $rc -ne 0 ] && { tail -3 $O/kt.err; exit 1; }
python3 $R/tools/prof_summary.py $O/kt $O/r03_rwkv7_b1 --skip-frac 0.0001; rm -rf $O/kt; head -30 $O/r03_rwkv7_b1_kernel_stats.csv
