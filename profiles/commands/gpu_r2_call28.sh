#!/bin/bash
# round-2 GPU call 28: rocprofv3 kernel trace + FETCH_SIZE / WRITE_SIZE passes of the 1536-token profile pass at 512 streams per lane
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2c28; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
PASS="--streams 512 --lanes 1 --lane-streams 512 --no-graph --no-events --step-sync --overlap 0 --sample-tokens 1536 --window 1536 --steps 1 --warmup 0 --cpu-sample 0 --profile-tokens 0"
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_kt -o kt -- python3 $R/bench.py $PASS > $O/kt.out 2> $O/kt.err; rc=$?
echo "kernel-trace rc=$rc"; [ $rc -eq 0 ] && python3 $R/tools/prof_summary.py /tmp/p_kt $O/r02c_pass1536_s512 && head -16 $O/r02c_pass1536_s512_kernel_stats.csv
[ $rc -ne 0 ] && { tail -5 $O/kt.err; exit 1; }
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 500 rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/p_$C -o pmc -- python3 $R/bench.py $PASS > $O/pmc_$C.out 2> $O/pmc_$C.err; rc=$?
  echo "pmc $C rc=$rc"
  [ $rc -eq 0 ] && python3 $R/tools/prof_summary.py /tmp/p_$C $O/r02c_pmc1536_$C
  [ $rc -ne 0 ] && { tail -5 $O/pmc_$C.err; exit 1; }
done
exit 0
