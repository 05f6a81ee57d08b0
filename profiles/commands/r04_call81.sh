#!/bin/bash
# round-4 GPU call 81: the whole GPU suite and smoke at the round's HEAD (last GPU call of the round)
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4c81; mkdir -p $O
cd $R
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1 || { tail -30 $O/pytest_gpu.log; exit 1; }
tail -2 $O/pytest_gpu.log
timeout -k 10 120 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1 || { tail -20 $O/smoke.log; exit 1; }
tail -1 $O/smoke.log
