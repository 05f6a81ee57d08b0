#!/bin/bash
# round-4 GPU call 76: the library rebuilt from clean by __graft_entry__.build(): smoke and the round-4 tests
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4c76; mkdir -p $O
cd $R
timeout -k 10 120 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1 || { tail -20 $O/smoke.log; exit 1; }
tail -1 $O/smoke.log
timeout -k 10 600 python3 -m pytest tests/test_gpu_round4.py tests/test_gpu_kernels.py -m gpu -x -q > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -1 $O/pytest.log
