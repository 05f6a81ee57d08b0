#!/bin/bash
# round-3 GPU call 31: rmsnorm_small_kernel with every global operand requested up-front (wave per row, no index divisions): parity, step times
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3c31; mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_round2.py tests/test_gpu_round3.py -m gpu -x -q -k "rmsnorm or wide or config5 or ctx1024" > $O/pytest.log 2>&1; rc=$?
tail -3 $O/pytest.log; echo "pytest rc=$rc"
[ $rc -ne 0 ] && exit 1
W="wm=256,wmode=3"
timeout -k 10 400 python tools/bench_decode_step.py --iters 60 --max-streams 1024 "b512:B=512,L=1,$W" "b512_2l:B=512,L=2,$W" "b1024:B=1024,L=1,$W" "b512_prof:B=512,L=1,$W,g=0,prof=1" > $O/steps.jsonl 2> $O/steps.err; cut -c1-420 $O/steps.jsonl
