#!/bin/bash
# round-2 GPU call 15: B = 1 step with / without the fused norm; default bench; kernel-trace + PMC passes of the profile-pass configuration; RWKV7 check
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2c15; mkdir -p $O
cd $R
timeout -k 10 200 python tools/bench_decode_step.py --iters 200 --max-streams 16 "b1:B=1,L=1,pos=767" "b1_fn:B=1,L=1,pos=767,fn=1" "b8:B=8,L=1,pos=767" "b8_fn:B=8,L=1,pos=767,fn=1" "b1_prof:B=1,L=1,pos=767,prof=1" > $O/b1.jsonl 2> $O/b1.err
python - <<'PY'
import json
for l in open('gpurun_out/r2c15/b1.jsonl'):
    d=json.loads(l); print(d['config'], d['ms_per_step_lane'], d['tokens_per_s_device_time'], d.get('families_us_per_step_lane0',''), d['errors'][:1])
PY
timeout -k 10 200 python tools/bench_rwkv7.py 384 1024 128 3 > $O/rwkv7.out 2> $O/rwkv7.err; tail -3 $O/rwkv7.out
timeout -k 10 700 python bench.py --steps 6 --warmup 2 > $O/bench_default_k6.json 2> $O/bench_default_k6.err; rc=$?
tail -3 $O/bench_default_k6.err; cut -c1-260 $O/bench_default_k6.json; echo; echo "bench rc=$rc"
[ $rc -ge 124 ] && exit 1
cd /tmp && export TMPDIR=/tmp
PASS="--streams 384 --lanes 1 --lane-streams 384 --no-graph --no-events --step-sync --sample-tokens 1536 --window 1536 --steps 1 --warmup 0 --cpu-sample 0 --profile-tokens 0"
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_kt -o kt -- python3 $R/bench.py $PASS > $O/kt.out 2> $O/kt.err; rc=$?
echo "kernel-trace rc=$rc"; [ $rc -eq 0 ] && python3 $R/tools/prof_summary.py /tmp/p_kt $O/r02_pass1536_s384 && head -14 $O/r02_pass1536_s384_kernel_stats.csv
[ $rc -ge 124 ] && exit 1
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 700 rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/p_$C -o pmc -- python3 $R/bench.py $PASS > $O/pmc_$C.out 2> $O/pmc_$C.err; rc=$?
  echo "pmc $C rc=$rc"
  [ $rc -eq 0 ] && python3 $R/tools/prof_summary.py /tmp/p_$C $O/r02_pmc1536_s384_$C
  [ $rc -ge 124 ] && exit 1
done
exit 0
