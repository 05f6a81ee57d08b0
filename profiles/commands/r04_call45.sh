#!/bin/bash
# round-4 GPU call 45: big-M GEMM 128 x 128 tiles with half K tiles staged between the wave's own MFMAs (PIPE): parity tests over the encode paths, then encode A/B with the knob gemm_pipe
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4c45; mkdir -p $O
cd $R
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q -k "encode or gemm or prefill or segment or batch or parity or kernels" > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
for rep in 1 2; do for v in 0 1; do
  CZ_BENCH_DEBUG_SET=gemm_pipe=$v timeout -k 10 300 python3 tools/encode_only.py 192 2048 2 > $O/enc.json 2>> $O/err.txt || { tail -3 $O/err.txt; exit 1; }
  python3 -c "
import json; d=json.load(open('$O/enc.json')); print('gemm_pipe $v', d['compress_tokens_per_s'], d['families_tflops'], d['families_ms']['gemm_bf16w_f32_mfma'])" | tee -a $O/pipe.txt
done; done
