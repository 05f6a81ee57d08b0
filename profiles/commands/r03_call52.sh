#!/bin/bash
# round-3 GPU call 52: rocprofv3 --kernel-trace of the bench's own pipeline (graph replay, two lanes + the encoder): per-kernel totals of the timed configuration itself
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3c52; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 700 rocprofv3 --kernel-trace --output-format csv -d $O/kt -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-sample 0 --profile-tokens 0 --no-events > $O/bench.json 2> $O/bench.err; rc=$?
echo "rc=$rc"; tail -2 $O/bench.err | cut -c1-200; cut -c1-160 $O/bench.json
[ $rc -ne 0 ] && exit 1
du -sh $O/kt | tail -1
python3 $R/tools/prof_summary.py $O/kt $O/r03_pipeline_k3 --skip-frac 0.0001 > /dev/null; rm -rf $O/kt; head -16 $O/r03_pipeline_k3_kernel_stats.csv
