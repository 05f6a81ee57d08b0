#!/bin/bash
# round-3 GPU call 14: gemm_mid with the ordered sum and the epilogue shared by the four waves of a block: parity, A/B of the wide step, phases
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3c14; mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_round2.py tests/test_gpu_round3.py -m gpu -x -q -k "wide_group or config5 or reprime_interval or rwkv7_wide" > $O/pytest.log 2>&1; rc=$?
tail -4 $O/pytest.log; echo "pytest rc=$rc"
[ $rc -ne 0 ] && exit 1
timeout -k 10 400 python tools/bench_decode_step.py --iters 60 --max-streams 1024 "q1_512:B=512,L=1,wm=256,wmode=3,mq=1" "q0_512:B=512,L=1,wm=256,wmode=3,mq=0" "q1_512_2l:B=512,L=2,wm=256,wmode=3,mq=1" "q0_512_2l:B=512,L=2,wm=256,wmode=3,mq=0" "q1_1024:B=1024,L=1,wm=256,wmode=3,mq=1" "q0_1024:B=1024,L=1,wm=256,wmode=3,mq=0" > $O/steps_quad.jsonl 2> $O/steps_quad.err; cut -c1-250 $O/steps_quad.jsonl
CANDLEZIP_AMD_LIB=$R/candlezip_amd/lib/libcandlezip_hip_middbg.so timeout -k 10 300 python tools/mid_phases.py 512 767 20 > $O/mid_phases_512.json 2> $O/mid_phases_512.err; python - <<PY
import json
d=json.load(open("$O/mid_phases_512.json"))
print(d["ms_per_step"])
for k,v in d["kernels"].items(): print(k, v)
PY
exit 0
