#!/bin/bash
# round-2 GPU call 31: the whole GPU suite + smoke at HEAD, then the driver-like bench line
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2c31; mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?
tail -4 $O/pytest.log; echo "pytest rc=$rc"
[ $rc -ne 0 ] && exit 1
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; rc=$?
tail -2 $O/smoke.log; echo "smoke rc=$rc"
[ $rc -ne 0 ] && exit 1
timeout -k 10 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_k20_w5.json 2> $O/bench_k20_w5.err; rc=$?
tail -3 $O/bench_k20_w5.err | cut -c1-300; cut -c1-400 $O/bench_k20_w5.json; echo; echo "bench rc=$rc"
exit 0
