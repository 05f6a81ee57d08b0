#!/bin/bash
# round-4 GPU call 53: bench lines (smollm K=6, rwkv7) after the attention diet
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4c53; mkdir -p $O
cd $R
timeout -k 10 300 python3 bench.py --steps 6 --warmup 2 > $O/bench_k6.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
python3 -c "
import json; d=json.load(open('$O/bench_k6.json')); print(d['value'], d['roofline']['frac'], d['roofline_mfma_gemm']['frac'], d['single_stream']['decompress_tokens_per_s'], d['compress_tokens_per_s_rank0'])"
timeout -k 10 300 python3 bench.py --backend rwkv7 > $O/bench_rwkv7.json 2> $O/bench_rwkv7.err || { tail -5 $O/bench_rwkv7.err; exit 1; }
python3 -c "
import json; d=json.load(open('$O/bench_rwkv7.json')); print(d['value'])"
