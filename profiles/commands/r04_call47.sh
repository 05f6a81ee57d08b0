#!/bin/bash
# round-4 GPU call 47: the device form of cz_expf (v_max / v_min / v_ldexp) against its definition on all 2^32 floats; parity suite; encode and wide step before / after
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4c47; mkdir -p $O
cd $R
timeout -k 10 300 python3 -m pytest tests/test_gpu_round4.py -m gpu -x -q -k "expf" > $O/pytest_expf.log 2>&1 || { tail -30 $O/pytest_expf.log; exit 1; }
tail -2 $O/pytest_expf.log
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1 || { tail -30 $O/pytest_gpu.log; exit 1; }
tail -2 $O/pytest_gpu.log
for i in 1 2; do timeout -k 10 300 python3 tools/encode_only.py 192 2048 2 > $O/enc$i.json 2>> $O/err.txt || { tail -3 $O/err.txt; exit 1; }; python3 -c "
import json; d=json.load(open('$O/enc$i.json')); print(d['compress_tokens_per_s'], d['families_tflops'], d['families_ms'])"; done
