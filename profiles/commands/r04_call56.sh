#!/bin/bash
# round-4 GPU call 56: PIPE half tiles unpadded + XOR-swizzled (3 x 128 x 64 blocks per CU): parity, encode, phase cycles
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4c56; mkdir -p $O
cd $R
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q -k "encode or gemm or prefill or segment or batch or parity or kernels or reprime or round_trip or rwkv" > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
for rep in 1 2; do
  timeout -k 10 300 python3 tools/encode_only.py 192 2048 2 > $O/enc.json 2>> $O/err.txt || { tail -3 $O/err.txt; exit 1; }
  python3 -c "
import json; d=json.load(open('$O/enc.json')); print(d['compress_tokens_per_s'], d['families_tflops'])" | tee -a $O/swz.txt
done
CANDLEZIP_AMD_LIB=$R/candlezip_amd/lib/libcandlezip_hip_gemmclock.so timeout -k 10 300 python3 tools/gemm_clock.py 192 2048 0 > $O/gemm_clock.jsonl 2>> $O/err.txt || { tail -3 $O/err.txt; exit 1; }
python3 -c "
import json
for l in open('$O/gemm_clock.jsonl'):
    d=json.loads(l); print(d['gemm_tflops_hip_events']); print(d['tiles_128x128_lm_head']); print(d['tiles_128x128_gate_up_swiglu']); print(d['tiles_128x64_qkv_o_down'])"
