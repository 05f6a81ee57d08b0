#!/bin/bash
# round-3 GPU call 2: as call 1 (whose first step failed on a wrong test expectation) + the RWKV7 sequence-parallel encode tests
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3c2; mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_round3.py tests/test_gpu_kernels.py -m gpu -x -q -k "round3 or attention" --durations=8 > $O/pytest_new.log 2>&1; rc=$?
tail -15 $O/pytest_new.log; echo "pytest new rc=$rc"
[ $rc -ne 0 ] && exit 1
timeout -k 10 1200 python -m pytest tests -m gpu -x -q --durations=10 > $O/pytest_all.log 2>&1; rc=$?
tail -15 $O/pytest_all.log; echo "pytest all rc=$rc"
[ $rc -ne 0 ] && exit 1
timeout -k 10 600 python bench.py --steps 6 --warmup 1 > $O/bench_default_k6.json 2> $O/bench_default_k6.err; rc=$?
tail -2 $O/bench_default_k6.err; cat $O/bench_default_k6.json | cut -c1-600; echo "bench rc=$rc"
[ $rc -ne 0 ] && exit 1
timeout -k 10 600 python bench.py --context 1024 --reprime-interval 1024 --window 1024 --steps 4 --warmup 1 --cpu-sample 0 --profile-tokens 3072 > $O/bench_ctx1024_r1024.json 2> $O/bench_ctx1024_r1024.err; rc=$?
tail -2 $O/bench_ctx1024_r1024.err; cat $O/bench_ctx1024_r1024.json | cut -c1-600; echo "bench ctx1024 rc=$rc"
[ $rc -ne 0 ] && exit 1
timeout -k 10 300 python tools/bench_rwkv7.py 384 1024 128 3 > $O/rwkv7_384.json 2> $O/rwkv7_384.err; rc=$?
tail -3 $O/rwkv7_384.err; cat $O/rwkv7_384.json | cut -c1-800; echo "rwkv7 bench rc=$rc"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_nosync -- python3 $R/bench.py --streams 128 --lanes 1 --lane-streams 128 --no-graph --no-events --overlap 0 --sample-tokens 1536 --window 1536 --steps 1 --warmup 0 --cpu-sample 0 --profile-tokens 0 > $O/pmc_nosync.out 2> $O/pmc_nosync.err; rc=$?
echo "pmc (no step-sync, single ROCm runtime) rc=$rc"; tail -5 $O/pmc_nosync.err | cut -c1-300
cd $R && python tools/prof_summary.py $O/pmc_nosync $O/pmc_nosync_summary 2>&1 | tail -2; rm -rf $O/pmc_nosync
exit 0
