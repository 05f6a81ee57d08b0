#!/bin/bash
# round-2 GPU call 14: whole GPU test suite, default bench (short K), single-file single-stream path (BASELINE config 2)
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2c14; mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?
tail -6 $O/pytest.log; echo "pytest rc=$rc"
[ $rc -ge 124 ] && exit 1
timeout -k 10 600 python bench.py --steps 5 --warmup 2 > $O/bench_default_k5.json 2> $O/bench_default_k5.err; rc=$?
tail -5 $O/bench_default_k5.err; cut -c1-700 $O/bench_default_k5.json; echo; echo "bench rc=$rc"
[ $rc -ge 124 ] && exit 1
timeout -k 10 400 python bench.py --streams 1 --lanes 1 --lane-streams 1 --sample-tokens 41933 --window 1024 --steps 41 --warmup 0 --cpu-sample 0 --profile-tokens 0 > $O/bench_single.json 2> $O/bench_single.err; rc=$?
tail -3 $O/bench_single.err; cut -c1-900 $O/bench_single.json; echo; echo "single rc=$rc"
exit 0
