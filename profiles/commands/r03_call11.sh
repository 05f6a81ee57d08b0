#!/bin/bash
# round-3 GPU call 11: split decode attention with every chunk's score in flight in the P.V kernel: parity of the split test, step times
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3c11; mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_round3.py -m gpu -x -q -k "split or config5" > $O/pytest.log 2>&1; rc=$?
tail -3 $O/pytest.log; echo "pytest rc=$rc"
[ $rc -ne 0 ] && exit 1
timeout -k 10 300 python tools/bench_decode_step.py --iters 200 --max-streams 16 "b1:B=1,L=1,g=1" "b4:B=4,L=1,g=1" "b8:B=8,L=1,g=1" "b14:B=14,L=1,g=1" > $O/steps_split.jsonl 2> $O/steps_split.err; cut -c1-230 $O/steps_split.jsonl
exit 0
