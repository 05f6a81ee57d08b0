#!/bin/bash
# round-3 GPU call 27: the RWKV7 profile pass at the bench's own group width (1 024 streams x 256 tokens) under rocprofv3 --kernel-trace
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3c27; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --output-format csv -d $O/kt -- python3 $R/tools/profile_pass.py rwkv7 --streams 1024 --tokens 256 > $O/kt.out 2> $O/kt.err; rc=$?; echo "rc=$rc"; grep "profile_pass\]" $O/kt.err | tail -1
[ $rc -ne 0 ] && exit 1
python3 $R/tools/prof_summary.py $O/kt $O/r03_passrwkv7_s1024_t256 --skip-frac 0.0001; rm -rf $O/kt; head -16 $O/r03_passrwkv7_s1024_t256_kernel_stats.csv
