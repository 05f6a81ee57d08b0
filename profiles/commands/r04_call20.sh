#!/bin/bash
# round-4 GPU call 20: the bench lines of the round at the head: the default line as the driver runs it (now with the `single_stream` object), the rwkv7 line, config 5's schedule
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4c20; mkdir -p $O
cd $R
timeout -k 10 560 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/r04_bench_driver_like_k20_w5.json 2> $O/bench.err; rc=$?
tail -3 $O/bench.err | cut -c1-200; [ $rc -ne 0 ] && exit 1
python3 -c "
import json;d=json.loads(open('$O/r04_bench_driver_like_k20_w5.json').read().strip().splitlines()[-1]);print('smollm', d['value'], d['roofline']['frac'], d.get('single_stream'), d['cpu_baseline']['value'])"
timeout -k 10 300 python3 bench.py --backend rwkv7 > $O/r04_bench_rwkv7.json 2> $O/bench_rwkv7.err; rc=$?
tail -2 $O/bench_rwkv7.err | cut -c1-200; [ $rc -ne 0 ] && exit 1
python3 -c "
import json;d=json.loads(open('$O/r04_bench_rwkv7.json').read().strip().splitlines()[-1]);print('rwkv7', d['value'], d['roofline']['frac'])"
