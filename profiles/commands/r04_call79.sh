#!/bin/bash
# round-4 GPU call 79: the re-prime's prefill attention with buffer-addressed transposed-K loads: decode parity tests, then the kernel trace of the profile pass (2 048 tokens) for its per-launch time
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4c79; mkdir -p $O
cd $R
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q -k "reprime or round_trip or decode or attention or kernels" > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -1 $O/pytest.log
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --output-format csv -d $O/kt -- python3 $R/tools/profile_pass.py smollm > $O/kt.out 2> $O/kt.err || { tail -5 $O/kt.err; exit 1; }
python3 $R/tools/prof_summary.py $O/kt $O/pass2048 --skip-frac 0.25; rm -rf $O/kt; grep "attention_v2\|dec_attn" $O/pass2048_kernel_stats.csv
