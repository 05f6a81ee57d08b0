#!/bin/bash
# round-4 GPU call 2: parity of the new gemm_mid kernels (consecutive chunks per wave; gate|up with unrolled rounds), then the wide decode step at 512
# streams with round 3's kernels (cpw=0) against the new ones (cpw=-1): graph replay, and eager with per-family events
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4c2; mkdir -p $O
cd $R && timeout -k 10 600 python3 -m pytest tests/test_gpu_round2.py tests/test_gpu_round3.py tests/test_gpu_round4.py -m gpu -x -q -k "wide_group or config5 or rwkv7_wide or knobs or several_streams" > $O/pytest.log 2>&1; rc=$?
tail -3 $O/pytest.log
[ $rc -ne 0 ] && exit 1
timeout -k 10 300 python3 tools/bench_decode_step.py --iters 60 --max-streams 512 \
  old:B=512,wm=256,wmode=3,cpw=0 new:B=512,wm=256,wmode=3,cpw=-1 old2:B=512,wm=256,wmode=3,cpw=0 new2:B=512,wm=256,wmode=3,cpw=-1 \
  oldp:B=512,wm=256,wmode=3,cpw=0,prof=1 newp:B=512,wm=256,wmode=3,cpw=-1,prof=1 \
  old_chain:B=512,wm=256,wmode=3,cpw=0,av=1 new_chain:B=512,wm=256,wmode=3,cpw=-1,av=1 > $O/steps.jsonl 2> $O/steps.err; rc=$?
cat $O/steps.jsonl | cut -c1-700; tail -3 $O/steps.err
exit $rc
