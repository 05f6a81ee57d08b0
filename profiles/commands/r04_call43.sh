#!/bin/bash
# round-4 GPU call 43: the whole GPU suite and the driver-like bench line at the buffer-addressed GEMMs
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4c43; mkdir -p $O
cd $R
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1 || { tail -30 $O/pytest_gpu.log; exit 1; }
tail -2 $O/pytest_gpu.log
timeout -k 10 120 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1 || { tail -20 $O/smoke.log; exit 1; }
tail -1 $O/smoke.log
timeout -k 10 300 python3 bench.py --steps 6 --warmup 2 > $O/bench_k6.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
python3 -c "
import json; d=json.load(open('$O/bench_k6.json')); print(d['value'], d['roofline']['frac'], d['roofline_mfma_gemm'], d['single_stream']['decompress_tokens_per_s'], d['compress_tokens_per_s_rank0'])"
