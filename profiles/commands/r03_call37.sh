#!/bin/bash
# round-3 GPU call 37: decode attention with the chunk-sum chains of a wave on 12 lanes side by side (were 3 lanes x 4 chunks in series): parity of every decode-attention form, A/B steps
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3c37; mkdir -p $O
cd $R
timeout -k 10 1000 python -m pytest tests/test_gpu_round2.py tests/test_gpu_round3.py tests/test_gpu_model.py tests/test_gpu_kernels.py -m gpu -x -q -k "not rwkv" > $O/pytest.log 2>&1; rc=$?
tail -3 $O/pytest.log; echo "pytest rc=$rc"
[ $rc -ne 0 ] && exit 1
W="wm=256,wmode=3"
CFG="attn_1l:B=512,L=1,av=2,$W both_1l:B=512,L=1,$W both_2l:B=512,L=2,$W b1:B=1,L=1 b128:B=128,L=1"
for rep in 1 2 3; do
timeout -k 10 300 python tools/bench_decode_step.py --iters 40 --max-streams 512 --sym 0 $CFG > $O/steps_new_$rep.jsonl 2> $O/steps_new.err || exit 1
python -c "
import json
print('new ', [ (json.loads(l)['config'][:8], json.loads(l)['ms_per_step_lane'][0]) for l in open('$O/steps_new_$rep.jsonl')])"
CANDLEZIP_AMD_LIB=$R/candlezip_amd/lib/libcandlezip_hip_prev.so timeout -k 10 300 python tools/bench_decode_step.py --iters 40 --max-streams 512 --sym 0 $CFG > $O/steps_prev_$rep.jsonl 2> $O/steps_prev.err || exit 1
python -c "
import json
print('prev', [ (json.loads(l)['config'][:8], json.loads(l)['ms_per_step_lane'][0]) for l in open('$O/steps_prev_$rep.jsonl')])"
done
