#!/bin/bash
# round-3 GPU call 44: soak: the random-schedule fuzz against the oracle with other seeds, 120 cases each
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3c44; mkdir -p $O
cd $R
for seed in 7 1234567; do
  CZ_FUZZ_SEED=$seed CZ_FUZZ_CASES=120 timeout -k 10 900 python -m pytest tests/test_gpu_fuzz.py -m gpu -x -q > $O/fuzz_$seed.log 2>&1; rc=$?
  tail -2 $O/fuzz_$seed.log; echo "seed $seed rc=$rc"
  [ $rc -ne 0 ] && { tail -30 $O/fuzz_$seed.log; exit 1; }
done
exit 0
