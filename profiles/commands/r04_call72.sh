#!/bin/bash
# round-4 GPU call 72: the driver-like bench line at the round's final state (four-wave decode-attention blocks on the two lanes; the profile pass in the same block shape)
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4c72; mkdir -p $O
cd $R
timeout -k 10 560 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/r04_bench_driver_like_k20_w5_final.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
python3 -c "
import json; d=json.load(open('$O/r04_bench_driver_like_k20_w5_final.json')); r=d['roofline']; print(d['value'], r['frac'], r.get('avg_launch_us'), r.get('avg_kernel_us_rocprof'), r.get('frac_rocprof'), r.get('traffic'), r.get('block_shape','')[:60], d['roofline_mfma_gemm']['frac'])"
