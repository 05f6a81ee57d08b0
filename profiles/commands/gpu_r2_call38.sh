#!/bin/bash
# round-2 GPU call 38: the last layer's tail on the wanted rows only (encode: the rows that predict a coded symbol; re-prime: the last row)
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2c38; mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?
tail -6 $O/pytest.log; echo "pytest rc=$rc"
[ $rc -ne 0 ] && exit 1
timeout -k 10 300 python tools/encode_only.py 192 2048 2 > $O/encode_only.json 2> $O/encode_only.err; rc=$?
cut -c1-600 $O/encode_only.json; echo "encode_only rc=$rc"
[ $rc -ne 0 ] && exit 1
timeout -k 10 900 python bench.py --steps 8 --warmup 2 --cpu-sample 0 > $O/bench_k8.json 2> $O/bench_k8.err; rc=$?
tail -2 $O/bench_k8.err | cut -c1-300; cut -c1-300 $O/bench_k8.json; echo; echo "bench rc=$rc"
exit 0
