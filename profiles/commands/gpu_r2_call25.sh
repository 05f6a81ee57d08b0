#!/bin/bash
# round-2 GPU call 25: attention_v2 with double-buffered tiles (one barrier per chunk and phase), 3 blocks per CU for the 512-key variant;
# clamped CDF logit loads; compress beside decompress
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2c25; mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?
tail -5 $O/pytest.log; echo "pytest rc=$rc"
[ $rc -ne 0 ] && exit 1
timeout -k 10 300 python tools/encode_only.py 192 2048 2 > $O/encode_only.json 2> $O/encode_only.err; rc=$?
cut -c1-600 $O/encode_only.json; echo "encode_only rc=$rc"
[ $rc -ne 0 ] && exit 1
for ov in 0 1; do
timeout -k 10 900 python bench.py --steps 6 --warmup 1 --overlap $ov --cpu-sample 0 > $O/bench_ov$ov.json 2> $O/bench_ov$ov.err; rc=$?
tail -2 $O/bench_ov$ov.err | cut -c1-300; cut -c1-300 $O/bench_ov$ov.json; echo; echo "bench overlap=$ov rc=$rc"
[ $rc -ne 0 ] && exit 1
done
exit 0
