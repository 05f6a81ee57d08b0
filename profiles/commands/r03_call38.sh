#!/bin/bash
# round-3 GPU call 38: cross_entropy_bits_over_span_rwkv on the device (sequence-parallel forward + literals-mode probability kernel) against the oracle; the rwkv and xe tests
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3c38; mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_round3.py -m gpu -x -q -k "rwkv or cross_entropy or xe" > $O/pytest.log 2>&1; rc=$?
tail -15 $O/pytest.log; echo "pytest rc=$rc"
exit $rc
