#!/bin/bash
# round-4 GPU call 10: the whole GPU suite + smoke at the head with the VALU projections of narrow groups, then the single-stream line (BASELINE config 2 as the
# reference CLI runs a file: ONE 41 933-token stream)
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4c10; mkdir -p $O
cd $R && timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?
tail -3 $O/pytest.log
[ $rc -ne 0 ] && exit 1
timeout -k 10 120 python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 || { tail -5 $O/smoke.log; exit 1; }
tail -1 $O/smoke.log
timeout -k 10 400 python3 bench.py --streams 1 --lanes 1 --lane-streams 1 --sample-tokens 41933 --window 1024 --steps 41 --warmup 0 --cpu-sample 0 --profile-tokens 0 > $O/bench_single_stream.json 2> $O/bench_single_stream.err; rc=$?
tail -1 $O/bench_single_stream.err; python3 -c "import json;d=json.loads(open('$O/bench_single_stream.json').read().strip().splitlines()[-1]);print('single stream', d['value'], d['compress_tokens_per_s_rank0'], d['decompress_tokens_per_s_rank0'], d['windows_verified'], d['generations_verified'])"
exit $rc
