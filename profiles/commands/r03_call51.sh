#!/bin/bash
# round-3 GPU call 51: big-M GEMM outputs stored non-temporally (M >= 8192): gemm parity, encode-only A/B against the previous library
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3c51; mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -m gpu -x -q -k "gemm or encode or payload" > $O/pytest.log 2>&1; rc=$?
tail -2 $O/pytest.log; echo "pytest rc=$rc"; [ $rc -ne 0 ] && exit 1
for rep in 1 2 3; do
for which in new prev; do
  if [ $which = prev ]; then export CANDLEZIP_AMD_LIB=$R/candlezip_amd/lib/libcandlezip_hip_prev.so; else unset CANDLEZIP_AMD_LIB; fi
  timeout -k 10 300 python tools/encode_only.py 128 2048 2 > $O/enc_${which}_$rep.json 2> $O/enc.err || exit 1
  python -c "
import json
d=json.loads(open('$O/enc_${which}_$rep.json').read()); print('$which', d['compress_tokens_per_s'], d['gemm_tflops'], d['attention_tflops'])"
done
done
