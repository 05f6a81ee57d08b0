#!/bin/bash
# round-2 GPU call 19: attention_v3 parity, encode alone with attention_v3 / attention_v2
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2c19; mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py tests/test_gpu_fuzz.py tests/test_gpu_checkpoints.py tests/test_gpu_round2.py -m gpu -x -q > $O/pytest.log 2>&1; rc=$?
tail -5 $O/pytest.log; echo "pytest rc=$rc"
[ $rc -ne 0 ] && exit 1
for v in 0 2 0 2; do timeout -k 10 200 python tools/encode_only.py 192 2048 2 $v | tee -a $O/encode_only.jsonl | cut -c1-400; done
exit 0
