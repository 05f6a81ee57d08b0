#!/bin/bash
# round-4 GPU call 75: ONE WHOLE generation at the final kernels (K = 34 windows = every one of the 1 024 samples compressed, decompressed and compared end to end), and config 5's schedule (context 1024 / reprime 1024)
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4c75; mkdir -p $O
cd $R
timeout -k 10 700 python3 bench.py --steps 34 --warmup 0 --cpu-sample 0 --profile-tokens 0 --single-stream-tokens 0 > $O/r04_bench_k34_one_whole_generation.json 2> $O/bench_k34.err || { tail -5 $O/bench_k34.err; exit 1; }
python3 -c "
import json; d=json.load(open('$O/r04_bench_k34_one_whole_generation.json')); print(d['value'], d['generations_verified'], d['windows_verified'])"
timeout -k 10 400 python3 bench.py --context 1024 --reprime-interval 1024 --window 1024 --steps 6 --warmup 2 --cpu-sample 0 --profile-tokens 0 --single-stream-tokens 0 > $O/r04_bench_ctx1024_r1024.json 2> $O/bench_ctx1024.err || { tail -5 $O/bench_ctx1024.err; exit 1; }
python3 -c "
import json; d=json.load(open('$O/r04_bench_ctx1024_r1024.json')); print(d['value'], d['roofline']['frac'])"
timeout -k 10 300 python3 bench.py --backend rwkv7 > $O/r04_bench_rwkv7_final.json 2> $O/bench_rwkv7.err || { tail -5 $O/bench_rwkv7.err; exit 1; }
python3 -c "
import json; d=json.load(open('$O/r04_bench_rwkv7_final.json')); print(d['value'])"
