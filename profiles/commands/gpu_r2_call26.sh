#!/bin/bash
# round-2 GPU call 26: gemm_mid sums on the round's last wave (tests + step microbench); attention_v2 with parts masked off (tuning build)
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2c26; mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_round2.py tests/test_gpu_model.py -m gpu -x -q > $O/pytest.log 2>&1; rc=$?
tail -3 $O/pytest.log; echo "pytest rc=$rc"
[ $rc -ne 0 ] && exit 1
timeout -k 10 600 python tools/bench_decode_step.py --max-streams 512 --iters 60 \
  "b512_nw4_prof:B=512,L=1,wm=256,wmode=3,prof=1,nw=4" "b512_nw4_l2:B=512,L=2,wm=256,wmode=3,nw=4" "b512_nw0_l2:B=512,L=2,wm=256,wmode=3,nw=0" "b512_nw8_l2:B=512,L=2,wm=256,wmode=3,nw=8" > $O/steps.jsonl 2> $O/steps.err; rc=$?
cut -c1-700 $O/steps.jsonl; echo "microbench rc=$rc"
[ $rc -ne 0 ] && exit 1
export CANDLEZIP_AMD_LIB=$R/candlezip_amd/lib/libcandlezip_hip_attndbg.so
for m in 0 1 2 4 8 16 32 6 24 31 63; do
CZ_ATTN_DBG=$m timeout -k 10 200 python tools/encode_only.py 96 2048 1 > $O/attn_dbg_$m.json 2> $O/attn_dbg_$m.err; rc=$?
echo "mask $m: $(cut -c1-330 $O/attn_dbg_$m.json)"
[ $rc -ne 0 ] && { tail -3 $O/attn_dbg_$m.err; exit 1; }
done
exit 0
