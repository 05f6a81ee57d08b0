#!/bin/bash
# round-2 GPU call 40: the full-size sample test on its own (timing), then the whole suite
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2c40; mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_round2.py -m gpu -x -q -k full_size --durations=3 > $O/pytest_full_size.log 2>&1; rc=$?
tail -8 $O/pytest_full_size.log; echo "pytest full_size rc=$rc"
[ $rc -ne 0 ] && exit 1
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?
tail -3 $O/pytest.log; echo "pytest rc=$rc"
exit 0
