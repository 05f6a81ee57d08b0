#!/bin/bash
# round-4 GPU call 30: the new threshold test (one group just past the wide-step thresholds, BASELINE shapes) + the whole GPU suite at the head
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4c30; mkdir -p $O
cd $R && timeout -k 10 1100 python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?
tail -3 $O/pytest.log
exit $rc
