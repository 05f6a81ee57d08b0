#!/bin/bash
# round-4 GPU call 69: the self-test line with four-wave decode-attention blocks (dec_attn_waves = 4) against eight, K = 6
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4c69; mkdir -p $O
cd $R
for v in 4 8; do
  timeout -k 10 400 python3 bench.py --steps 6 --warmup 2 --cpu-sample 0 --single-stream-tokens 0 --debug-set dec_attn_waves=$v > $O/bench_aw$v.json 2> $O/bench_aw$v.err || { tail -5 $O/bench_aw$v.err; exit 1; }
  python3 -c "
import json; d=json.load(open('$O/bench_aw$v.json')); print('dec_attn_waves $v', d['value'], d['roofline']['frac'], d['roofline'].get('avg_launch_us'))"
done
