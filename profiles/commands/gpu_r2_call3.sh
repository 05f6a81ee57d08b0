#!/bin/bash
# round-2 GPU call 3: decode-step microbenchmarks (lanes / XCD-masked lanes / attention variant), then ONE PMC pass with per-step synchronisation
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2c3; mkdir -p $O
cd $R
timeout -k 10 500 python tools/bench_decode_step.py --iters 60 \
  "eager_prof:B=128,L=1,pos=767,prof=1" \
  "g1:B=128,L=1,pos=767" "g1_av1:B=128,L=1,pos=767,av=1" \
  "g1_p575:B=128,L=1,pos=575" "g1_p575_av1:B=128,L=1,pos=575,av=1" "g1_p1023:B=128,L=1,pos=1023" "g1_p1023_av1:B=128,L=1,pos=1023,av=1" \
  "g3:B=128,L=3,pos=767" "g3_av1:B=128,L=3,pos=767,av=1" \
  "x8_b48:B=48,L=8,pos=767,x=1" "x8_b64:B=64,L=8,pos=767,x=1" "n8_b48:B=48,L=8,pos=767,x=0" \
  "x4_b96:B=96,L=4,pos=767,x=1" "x3_b128:B=128,L=3,pos=767,x=1" "x1_b48:B=48,L=1,pos=767,x=1" "n1_b48:B=48,L=1,pos=767" \
  > $O/steps.jsonl 2> $O/steps.err; rc=$?
cat $O/steps.jsonl | cut -c1-600; tail -3 $O/steps.err; echo "microbench rc=$rc"
[ $rc -ge 124 ] && exit 1
cd /tmp && export TMPDIR=/tmp
PASS="--streams 128 --lanes 1 --no-graph --no-events --step-sync --sample-tokens 1536 --window 1536 --steps 1 --warmup 0 --cpu-sample 0 --profile-tokens 0"
PYTHONFAULTHANDLER=1 timeout -k 10 700 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/p_F -o pmc -- python3 $R/bench.py $PASS > $O/pmc_F.out 2> $O/pmc_F.err
rc=$?; echo "pmc FETCH_SIZE (step-sync) rc=$rc"; tail -4 $O/pmc_F.err
if [ $rc -eq 0 ]; then
  python3 $R/tools/prof_summary.py /tmp/p_F $O/r02_pmc1536_FETCH_SIZE
  timeout -k 10 700 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/p_W -o pmc -- python3 $R/bench.py $PASS > $O/pmc_W.out 2> $O/pmc_W.err
  rc=$?; echo "pmc WRITE_SIZE rc=$rc"
  [ $rc -eq 0 ] && python3 $R/tools/prof_summary.py /tmp/p_W $O/r02_pmc1536_WRITE_SIZE
fi
exit 0
