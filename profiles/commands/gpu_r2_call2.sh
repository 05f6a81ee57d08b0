#!/bin/bash
# round-2 GPU call 2: the GPU test suite, a short run of the new persistent-stream bench, one unrestricted PMC pass without batched submissions
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2c2; mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?
tail -15 $O/pytest.log; echo "pytest rc=$rc"
[ $rc -ge 124 ] && exit 1
timeout -k 10 600 python bench.py --steps 3 --warmup 1 --cpu-sample 0 > $O/bench_k3.json 2> $O/bench_k3.err; rc=$?
tail -8 $O/bench_k3.err; cat $O/bench_k3.json | cut -c1-1500; echo "bench rc=$rc"
[ $rc -ge 124 ] && exit 1
cd /tmp && export TMPDIR=/tmp
PASS="--streams 128 --lanes 1 --no-graph --no-events --sample-tokens 1536 --window 1536 --steps 1 --warmup 0 --cpu-sample 0 --profile-tokens 0"
CZ_DUMP_MAPS=$O/maps_pmc.txt PYTHONFAULTHANDLER=1 timeout -k 10 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/p_F -o pmc -- python3 $R/bench.py $PASS > $O/pmc_F.out 2> $O/pmc_F.err
rc=$?; echo "pmc FETCH_SIZE (no events, no graph) rc=$rc"; tail -5 $O/pmc_F.err
if [ $rc -eq 0 ]; then
  python3 $R/tools/prof_summary.py /tmp/p_F $O/r02_pmc1536_FETCH_SIZE
  timeout -k 10 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/p_W -o pmc -- python3 $R/bench.py $PASS > $O/pmc_W.out 2> $O/pmc_W.err
  rc=$?; echo "pmc WRITE_SIZE rc=$rc"
  [ $rc -eq 0 ] && python3 $R/tools/prof_summary.py /tmp/p_W $O/r02_pmc1536_WRITE_SIZE
fi
exit 0
