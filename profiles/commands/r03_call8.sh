#!/bin/bash
# round-3 GPU call 8: phase stamps inside the narrow-group decode kernels (B = 1, 8)
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3c8; mkdir -p $O
cd $R
export CANDLEZIP_AMD_LIB=$R/candlezip_amd/lib/libcandlezip_hip_decdbg.so
timeout -k 10 200 python tools/dec_phases.py 1 767 100 > $O/dec_phases_b1.json 2> $O/dec_phases_b1.err; cat $O/dec_phases_b1.json; tail -2 $O/dec_phases_b1.err
timeout -k 10 200 python tools/dec_phases.py 8 767 100 > $O/dec_phases_b8.json 2> $O/dec_phases_b8.err; cat $O/dec_phases_b8.json | head -60
exit 0
