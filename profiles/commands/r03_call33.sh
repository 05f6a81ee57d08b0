#!/bin/bash
# round-3 GPU call 33: whole-device decode attention with its first K round requested together with Q (against the build without): parity, attention-only and full steps
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3c33; mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_round2.py tests/test_gpu_round3.py -m gpu -x -q -k "wide or config5 or ctx1024" > $O/pytest.log 2>&1; rc=$?
tail -3 $O/pytest.log; echo "pytest rc=$rc"
[ $rc -ne 0 ] && exit 1
W="wm=256,wmode=3"
CFG="attn_1l:B=512,L=1,av=2,$W attn_1l_1023:B=512,L=1,av=2,pos=1023,$W both_1l:B=512,L=1,$W both_2l:B=512,L=2,$W"
for rep in 1 2 3; do
timeout -k 10 300 python tools/bench_decode_step.py --iters 40 --max-streams 512 --sym 0 $CFG > $O/steps_kpre1_$rep.jsonl 2> $O/steps_kpre1.err || exit 1
python -c "
import json
print('kpre1', [ (json.loads(l)['config'][:12], json.loads(l)['ms_per_step_lane'][0]) for l in open('$O/steps_kpre1_$rep.jsonl')])"
CANDLEZIP_AMD_LIB=$R/candlezip_amd/lib/libcandlezip_hip_dakpre0.so timeout -k 10 300 python tools/bench_decode_step.py --iters 40 --max-streams 512 --sym 0 $CFG > $O/steps_kpre0_$rep.jsonl 2> $O/steps_kpre0.err || exit 1
python -c "
import json
print('kpre0', [ (json.loads(l)['config'][:12], json.loads(l)['ms_per_step_lane'][0]) for l in open('$O/steps_kpre0_$rep.jsonl')])"
done
