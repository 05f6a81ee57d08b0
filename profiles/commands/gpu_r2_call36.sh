#!/bin/bash
# round-2 GPU call 36: 2048 streams per GPU (2 lanes x 1024, 193 GB of KV cache): does the wider group pay?
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2c36; mkdir -p $O
cd $R
timeout -k 10 900 python bench.py --streams 2048 --lanes 2 --lane-streams 1024 --steps 4 --warmup 1 --cpu-sample 0 --profile-tokens 0 > $O/bench_s2048.json 2> $O/bench_s2048.err; rc=$?
tail -3 $O/bench_s2048.err | cut -c1-300; cut -c1-300 $O/bench_s2048.json; echo; echo "bench rc=$rc"; rocm-smi --showmeminfo vram 2>/dev/null | tail -3
exit 0
