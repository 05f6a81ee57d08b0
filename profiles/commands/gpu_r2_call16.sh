#!/bin/bash
# round-2 GPU call 16: parity with the 16-wave decode attention; B = 1 step; single-file path again; RWKV7 group shapes
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2c16; mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?
tail -5 $O/pytest.log; echo "pytest rc=$rc"
[ $rc -ne 0 ] && exit 1
timeout -k 10 200 python tools/bench_decode_step.py --iters 200 --max-streams 64 "b1:B=1,L=1,pos=767" "b1_fn:B=1,L=1,pos=767,fn=1" "b8:B=8,L=1,pos=767" "b32:B=32,L=1,pos=767" "b64:B=64,L=1,pos=767" "b1_prof:B=1,L=1,pos=767,prof=1" > $O/b1.jsonl 2> $O/b1.err
python - <<'PY'
import json
for l in open('gpurun_out/r2c16/b1.jsonl'):
    d=json.loads(l); print(d['config'], d['ms_per_step_lane'], d['tokens_per_s_device_time'], d.get('families_us_per_step_lane0',''), d['errors'][:1])
PY
timeout -k 10 400 python bench.py --streams 1 --lanes 1 --lane-streams 1 --sample-tokens 41933 --window 1024 --steps 41 --warmup 0 --cpu-sample 0 --profile-tokens 0 > $O/bench_single.json 2> $O/bench_single.err; rc=$?
tail -1 $O/bench_single.err; cut -c1-200 $O/bench_single.json; echo
for cfg in "768 512 384 2" "768 512 768 1" "1024 512 512 2" "768 512 256 3"; do
  timeout -k 10 300 python tools/bench_rwkv7.py $cfg >> $O/rwkv7.out 2>> $O/rwkv7.err; tail -1 $O/rwkv7.out | cut -c1-330
done
exit 0
