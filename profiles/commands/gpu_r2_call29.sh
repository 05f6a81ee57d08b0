#!/bin/bash
# round-2 GPU call 29: lane shapes with this round's kernels (3 x 341, 4 x 256 against 2 x 512), RWKV7 shape after the CDF changes
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2c29; mkdir -p $O
cd $R
timeout -k 10 300 python tools/bench_rwkv7.py 384 1024 128 3 > $O/rwkv7.json 2> $O/rwkv7.err; rc=$?
cut -c1-500 $O/rwkv7.json; echo "rwkv7 rc=$rc"
[ $rc -ne 0 ] && { tail -3 $O/rwkv7.err; exit 1; }
for shape in "3 341 1023" "4 256 1024"; do
set -- $shape
timeout -k 10 600 python bench.py --steps 5 --warmup 1 --lanes $1 --lane-streams $2 --streams $3 --cpu-sample 0 --profile-tokens 0 > $O/bench_l$1x$2.json 2> $O/bench_l$1x$2.err; rc=$?
tail -1 $O/bench_l$1x$2.err | cut -c1-200; cut -c1-260 $O/bench_l$1x$2.json; echo; echo "bench $1x$2 rc=$rc"
[ $rc -ne 0 ] && exit 1
done
exit 0
