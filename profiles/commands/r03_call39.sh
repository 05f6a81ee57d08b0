#!/bin/bash
# round-3 GPU call 39: the line as the driver runs it and the rwkv7 line, with the final bench.py (rooflines that were not measured are null; no per-launch rocprof figure for the multi-shape GEMM)
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3c39; mkdir -p $O
cd $R
timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_like.json 2> $O/bench_driver_like.err; rc=$?
tail -1 $O/bench_driver_like.err; cut -c1-200 $O/bench_driver_like.json; echo "driver-like rc=$rc"; [ $rc -ne 0 ] && exit 1
timeout -k 10 600 python bench.py --backend rwkv7 --steps 2 --warmup 1 > $O/bench_rwkv7.json 2> $O/bench_rwkv7.err; rc=$?
tail -1 $O/bench_rwkv7.err; cut -c1-200 $O/bench_rwkv7.json; echo "rwkv7 rc=$rc"
exit $rc
