#!/bin/bash
# round-3 GPU call 10: split decode attention of small groups: parity, step times at B = 1 / 8 / 16 with the split form on and off, kernel trace
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3c10; mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_round3.py tests/test_gpu_model.py tests/test_gpu_round2.py tests/test_gpu_fuzz.py -m gpu -x -q > $O/pytest.log 2>&1; rc=$?
tail -6 $O/pytest.log; echo "pytest rc=$rc"
[ $rc -ne 0 ] && exit 1
timeout -k 10 300 python tools/bench_decode_step.py --iters 200 --max-streams 16 "b1:B=1,L=1,g=1" "b4:B=4,L=1,g=1" "b8:B=8,L=1,g=1" "b14:B=14,L=1,g=1" "b16:B=16,L=1,g=1" > $O/steps_split.jsonl 2> $O/steps_split.err; cut -c1-230 $O/steps_split.jsonl
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/b1_trace -- python3 $R/tools/bench_decode_step.py --iters 20 --max-streams 1 "b1:B=1,L=1,g=1" > $O/b1_trace.out 2> $O/b1_trace.err
python3 $R/tools/trace_gaps.py $O/b1_trace --last 2150 > $O/b1_gaps.txt 2>&1; head -16 $O/b1_gaps.txt; rm -rf $O/b1_trace
exit 0
