#!/bin/bash
# round-4 GPU call 19: soak of the random-schedule fuzz against the oracle with the VALU projections in the default path (groups of <= 16 streams take them): two other seeds, 120 cases each
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4c19; mkdir -p $O
cd $R
for seed in 20261004 424242; do
  CZ_FUZZ_SEED=$seed CZ_FUZZ_CASES=120 timeout -k 10 900 python3 -m pytest tests/test_gpu_fuzz.py -m gpu -x -q > $O/fuzz_$seed.log 2>&1; rc=$?
  tail -2 $O/fuzz_$seed.log; echo "seed $seed rc=$rc"
  [ $rc -ne 0 ] && { tail -30 $O/fuzz_$seed.log; exit 1; }
done
exit 0
