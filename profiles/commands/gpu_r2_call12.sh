#!/bin/bash
# round-2 GPU call 12: exact per-kernel durations (rocprofv3 kernel trace) of the eager decode step at B = 768 for the four wide-step forms
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2c12; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for cfg in "tiled:wm=1,wmode=1" "hybrid:wm=1,wmode=2" "mid:wm=1,wmode=3" "np2:wm=0"; do
  tag=${cfg%%:*}
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$tag -o t -- python3 $R/tools/bench_decode_step.py --iters 6 --max-streams 768 "$tag:B=768,L=1,g=0,${cfg#*:}" > $O/$tag.out 2> $O/$tag.err; rc=$?
  echo "== $tag rc=$rc"; python3 $R/tools/trace_step_kernels.py /tmp/tr_$tag | tee $O/$tag.kernels.txt
  [ $rc -ge 124 ] && exit 1
done
exit 0
