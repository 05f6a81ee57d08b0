#!/bin/bash
# round-3 GPU call 49: the RWKV7 fuzz (every encode / decode form, lanes, literals, span scorer) at its default and as a soak with two other seeds x 60 cases
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3c49; mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_fuzz.py -m gpu -x -q -k rwkv > $O/fuzz_default.log 2>&1; rc=$?; tail -2 $O/fuzz_default.log; [ $rc -ne 0 ] && { tail -30 $O/fuzz_default.log; exit 1; }
for seed in 5 424242; do
  CZ_FUZZ_RWKV_SEED=$seed CZ_FUZZ_RWKV_CASES=60 timeout -k 10 900 python -m pytest tests/test_gpu_fuzz.py -m gpu -x -q -k rwkv > $O/fuzz_$seed.log 2>&1; rc=$?
  tail -2 $O/fuzz_$seed.log; echo "seed $seed rc=$rc"
  [ $rc -ne 0 ] && { tail -30 $O/fuzz_$seed.log; exit 1; }
done
exit 0
