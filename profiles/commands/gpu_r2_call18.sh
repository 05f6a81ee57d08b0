#!/bin/bash
# round-2 GPU call 18: the suite once more at HEAD, then the bench exactly as the driver runs it
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2c18; mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?
tail -4 $O/pytest.log; echo "pytest rc=$rc"
[ $rc -ne 0 ] && exit 1
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout -k 10 1000 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_like.json 2> $O/bench_driver_like.err; rc=$?
tail -4 $O/bench_driver_like.err; cut -c1-330 $O/bench_driver_like.json; echo; echo "bench rc=$rc"
exit 0
