#!/bin/bash
# round-3 GPU call 29: fewer big-M GEMM blocks per CU (gemm_lds_kb: 84 = one 128x128 block / two 128x64 blocks instead of two / three), so that a decode-attention block
# fits beside them on every CU: does the encoder then run BESIDE the decode lanes?  Short pipeline runs.
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3c29; mkdir -p $O
cd $R
for kb in 0 84 56; do
  timeout -k 10 400 python bench.py --steps 3 --warmup 1 --cpu-sample 0 --profile-tokens 0 --debug-set gemm_lds_kb=$kb > $O/bench_kb$kb.json 2> $O/bench_kb$kb.err || { tail -3 $O/bench_kb$kb.err; exit 1; }
  python -c "import json;d=json.loads(open('$O/bench_kb$kb.json').read().strip().splitlines()[-1]);print('gemm_lds_kb=$kb', d['value'], 'compress', d['compress_tokens_per_s_rank0'], 'decompress', d['decompress_tokens_per_s_rank0'], d.get('roofline_mfma_gemm',{}).get('frac'))"
done
