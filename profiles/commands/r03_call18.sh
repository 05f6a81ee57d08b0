#!/bin/bash
# round-3 GPU call 18: gate|up gemm_mid at three blocks per CU (164 VGPRs under a 3-waves-per-SIMD bound: its 768 blocks at 512 rows in ONE generation)
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3c18; mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_round2.py -m gpu -x -q -k "wide_group and mid" > $O/pytest.log 2>&1; rc=$?
tail -3 $O/pytest.log; echo "pytest rc=$rc"
[ $rc -ne 0 ] && exit 1
timeout -k 10 400 python tools/bench_decode_step.py --iters 60 --max-streams 1024 "b512:B=512,L=1,wm=256,wmode=3" "b512_2l:B=512,L=2,wm=256,wmode=3" "b1024:B=1024,L=1,wm=256,wmode=3" "b512_prof:B=512,L=1,wm=256,wmode=3,g=0,prof=1" > $O/steps.jsonl 2> $O/steps.err; cut -c1-420 $O/steps.jsonl
exit 0
