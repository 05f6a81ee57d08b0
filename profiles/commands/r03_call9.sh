#!/bin/bash
# round-3 GPU call 9: after the attention V prefetch (one-block-per-query-head shape) and the hoisted norm weights: parity, step times, phases
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3c9; mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_round2.py tests/test_gpu_round3.py tests/test_gpu_fuzz.py -m gpu -x -q > $O/pytest.log 2>&1; rc=$?
tail -4 $O/pytest.log; echo "pytest rc=$rc"
[ $rc -ne 0 ] && exit 1
timeout -k 10 300 python tools/bench_decode_step.py --iters 200 --max-streams 16 "b1:B=1,L=1,g=1" "b8:B=8,L=1,g=1" "b16:B=16,L=1,g=1" > $O/b1_steps.jsonl 2> $O/b1_steps.err; cut -c1-330 $O/b1_steps.jsonl
CANDLEZIP_AMD_LIB=$R/candlezip_amd/lib/libcandlezip_hip_decdbg.so timeout -k 10 200 python tools/dec_phases.py 1 767 100 > $O/dec_phases_b1.json 2> $O/dec_phases_b1.err; cat $O/dec_phases_b1.json
exit 0
