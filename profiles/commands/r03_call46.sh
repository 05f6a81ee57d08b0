#!/bin/bash
# round-3 GPU call 46: logits written (lm_head GEMM) and read (CDF kernels) non-temporally: parity, decode step and encode-only A/B against the previous library
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3c46; mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py tests/test_gpu_round2.py -m gpu -x -q -k "cdf or gemm or payload or encode or decode_symbol or rwkv or wide" > $O/pytest.log 2>&1; rc=$?
tail -2 $O/pytest.log; echo "pytest rc=$rc"; [ $rc -ne 0 ] && exit 1
W="wm=256,wmode=3"
for rep in 1 2; do
for which in new prev; do
  if [ $which = prev ]; then export CANDLEZIP_AMD_LIB=$R/candlezip_amd/lib/libcandlezip_hip_prev.so; else unset CANDLEZIP_AMD_LIB; fi
  timeout -k 10 300 python tools/bench_decode_step.py --iters 40 --max-streams 512 "b512:B=512,L=1,$W" "b512_2l:B=512,L=2,$W" > $O/steps_${which}_$rep.jsonl 2> $O/steps.err || exit 1
  timeout -k 10 300 python tools/encode_only.py 96 2048 2 > $O/enc_${which}_$rep.json 2> $O/enc.err || exit 1
  python -c "
import json
print('$which', [ (json.loads(l)['config'][:8], json.loads(l)['ms_per_step_lane'][0]) for l in open('$O/steps_${which}_$rep.jsonl')], json.loads(open('$O/enc_${which}_$rep.json').read())['compress_tokens_per_s'])"
done
done
