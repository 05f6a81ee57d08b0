#!/bin/bash
# round-2 GPU call 21: fed decode (tests), the bench with the two-stage pipeline on the same streams
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2c21; mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?
tail -5 $O/pytest.log; echo "pytest rc=$rc"
[ $rc -ne 0 ] && exit 1
timeout -k 10 900 python bench.py --steps 8 --warmup 2 > $O/bench_k8.json 2> $O/bench_k8.err; rc=$?
tail -4 $O/bench_k8.err | cut -c1-300; cut -c1-300 $O/bench_k8.json; echo; echo "bench rc=$rc"
exit 0
