#!/bin/bash
# round-4 GPU call 70: four-wave decode-attention blocks where lanes share the device (auto): the whole GPU suite incl. the new block-shape test, then the self-test line (K = 6) and the driver-like line
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4c70; mkdir -p $O
cd $R
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1 || { tail -30 $O/pytest_gpu.log; exit 1; }
tail -2 $O/pytest_gpu.log
timeout -k 10 120 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1 || { tail -20 $O/smoke.log; exit 1; }
tail -1 $O/smoke.log
timeout -k 10 560 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/r04_bench_driver_like_k20_w5_final.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
python3 -c "
import json; d=json.load(open('$O/r04_bench_driver_like_k20_w5_final.json')); print(d['value'], d['roofline']['frac'], d['roofline'].get('avg_launch_us'), d['roofline_mfma_gemm']['frac'], d['single_stream']['decompress_tokens_per_s'])"
