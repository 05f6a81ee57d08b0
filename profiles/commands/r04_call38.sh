#!/bin/bash
# round-4 GPU call 38: big-M GEMM with buffer-descriptor addressing (scalar row offsets, one vector instruction per load / store): parity tests that walk it,
# phase cycles (tuning build), encode throughput; plus the VALU-in-the-shadow-of-MFMA microbenchmark
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4c38; mkdir -p $O
cd $R
timeout -k 10 200 candlezip_amd/lib/mfma_peak > $O/mfma_peak.jsonl 2> $O/err.txt || { tail -3 $O/err.txt; exit 1; }
tail -3 $O/mfma_peak.jsonl
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q -k "encode or gemm or prefill or segment or batch or parity" > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
CANDLEZIP_AMD_LIB=$R/candlezip_amd/lib/libcandlezip_hip_gemmclock.so timeout -k 10 300 python3 tools/gemm_clock.py 192 2048 0 > $O/gemm_clock.jsonl 2>> $O/err.txt || { tail -3 $O/err.txt; exit 1; }
cat $O/gemm_clock.jsonl
for i in 1 2; do timeout -k 10 300 python3 tools/encode_only.py 192 2048 2 > $O/enc$i.json 2>> $O/err.txt || { tail -3 $O/err.txt; exit 1; }; python3 -c "
import json; d=json.load(open('$O/enc$i.json')); print(d['compress_tokens_per_s'], d['families_tflops'])"; done
