#!/bin/bash
# round-2 GPU call 30: stream priorities for the encoder beside the decode lanes
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2c30; mkdir -p $O
cd $R
for pr in "1 0" "1 -1" "0 -1"; do
set -- $pr
timeout -k 10 600 python bench.py --steps 5 --warmup 1 --enc-priority $1 --dec-priority $2 --cpu-sample 0 --profile-tokens 0 > $O/bench_e$1_d$2.json 2> $O/bench_e$1_d$2.err; rc=$?
tail -1 $O/bench_e$1_d$2.err | cut -c1-200; cut -c1-260 $O/bench_e$1_d$2.json; echo; echo "bench enc=$1 dec=$2 rc=$rc"
[ $rc -ne 0 ] && { tail -5 $O/bench_e$1_d$2.err; exit 1; }
done
exit 0
