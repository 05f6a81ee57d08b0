#!/bin/bash
# round-3 GPU call 43: the whole GPU suite + smoke at the head with the staged literals-mode CDF and the RWKV span scorer
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3c43; mkdir -p $O
cd $R
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?
tail -4 $O/pytest.log; echo "pytest rc=$rc"
[ $rc -ne 0 ] && exit 1
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; rc=$?; tail -2 $O/smoke.log
exit $rc
