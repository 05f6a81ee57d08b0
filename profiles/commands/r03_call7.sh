#!/bin/bash
# round-3 GPU call 7: full GPU suite at the current head; the floor under a chain of dependent kernels; smollm group shapes (1 x 1024 against 2 x 512);
# the default lines of both back-ends
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3c7; mkdir -p $O
cd $R
timeout -k 10 1200 python -m pytest tests -m gpu -x -q --durations=5 > $O/pytest_all.log 2>&1; rc=$?
tail -10 $O/pytest_all.log; echo "pytest all rc=$rc"
[ $rc -ne 0 ] && exit 1
timeout -k 10 200 python tools/probe_chain.py > $O/probe_chain.jsonl 2> $O/probe_chain.err; cat $O/probe_chain.jsonl
timeout -k 10 500 python bench.py --steps 4 --warmup 1 --cpu-sample 0 --profile-tokens 0 --lanes 1 --lane-streams 1024 > $O/bench_1x1024.json 2> $O/bench_1x1024.err
python -c "import json;d=json.loads(open('$O/bench_1x1024.json').read().strip().splitlines()[-1]);print('1x1024', d['value'], d['compress_tokens_per_s_rank0'], d['decompress_tokens_per_s_rank0'])"
timeout -k 10 600 python bench.py --backend rwkv7 --steps 2 --warmup 1 > $O/bench_rwkv7.json 2> $O/bench_rwkv7.err; rc=$?
tail -1 $O/bench_rwkv7.err; cut -c1-260 $O/bench_rwkv7.json; echo "bench rwkv7 rc=$rc"
timeout -k 10 700 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_like_k20_w5.json 2> $O/bench_driver_like_k20_w5.err; rc=$?
tail -1 $O/bench_driver_like_k20_w5.err; cut -c1-260 $O/bench_driver_like_k20_w5.json; echo "bench driver-like rc=$rc"
exit 0
