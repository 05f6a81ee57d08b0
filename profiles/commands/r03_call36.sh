#!/bin/bash
# round-3 GPU call 36: the bench lines at the final head: as the driver runs it (K = 20, W = 5), config 5's schedule, and one whole generation (K = 34: every sample's round trip verified)
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3c36; mkdir -p $O
cd $R
timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_like.json 2> $O/bench_driver_like.err; rc=$?
tail -1 $O/bench_driver_like.err; cut -c1-200 $O/bench_driver_like.json; echo "driver-like rc=$rc"; [ $rc -ne 0 ] && exit 1
timeout -k 10 600 python bench.py --context 1024 --reprime-interval 1024 --window 1024 --steps 4 --warmup 1 --cpu-sample 0 --profile-tokens 3072 > $O/bench_ctx1024.json 2> $O/bench_ctx1024.err; rc=$?
tail -1 $O/bench_ctx1024.err; cut -c1-200 $O/bench_ctx1024.json; echo "ctx1024 rc=$rc"; [ $rc -ne 0 ] && exit 1
timeout -k 10 700 python bench.py --steps 34 --warmup 0 --cpu-sample 0 --profile-tokens 0 > $O/bench_k34.json 2> $O/bench_k34.err; rc=$?
tail -1 $O/bench_k34.err; cut -c1-200 $O/bench_k34.json; echo "k34 rc=$rc"
python -c "
import json
for n in ('driver_like','ctx1024','k34'):
    d=json.loads(open('$O/bench_'+n+'.json').read().strip().splitlines()[-1]); print(n, d['value'], d.get('windows_verified'), d.get('generations_verified'), d['roofline']['frac'], d['roofline'].get('frac_rocprof'))"
exit $rc
