#!/bin/bash
# round-3 GPU call 17: full GPU suite + smoke at the head, the driver-like headline line, config 5's line, the rwkv7 line
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3c17; mkdir -p $O
cd $R
timeout -k 10 1200 python -m pytest tests -m gpu -x -q --durations=5 > $O/pytest_all.log 2>&1; rc=$?
tail -9 $O/pytest_all.log; echo "pytest all rc=$rc"
[ $rc -ne 0 ] && exit 1
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout -k 10 700 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_like_k20_w5.json 2> $O/bench_driver_like_k20_w5.err; rc=$?
tail -1 $O/bench_driver_like_k20_w5.err; cut -c1-260 $O/bench_driver_like_k20_w5.json; echo "bench driver-like rc=$rc"
timeout -k 10 600 python bench.py --context 1024 --reprime-interval 1024 --window 1024 --steps 4 --warmup 1 --cpu-sample 0 --profile-tokens 3072 > $O/bench_ctx1024_r1024.json 2> $O/bench_ctx1024_r1024.err; rc=$?
cut -c1-200 $O/bench_ctx1024_r1024.json; echo "bench ctx1024 rc=$rc"
timeout -k 10 600 python bench.py --backend rwkv7 --steps 2 --warmup 1 > $O/bench_rwkv7.json 2> $O/bench_rwkv7.err; rc=$?
cut -c1-200 $O/bench_rwkv7.json; echo "bench rwkv7 rc=$rc"
exit 0
