#!/bin/bash
# round-3 GPU call 12: rw_state_kernel occupancy sweep (extra LDS per block -> fewer waves per CU -> several generations whose load / compute /
# store phases overlap), at 256 and 1024 streams
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3c12; mkdir -p $O
cd $R
for S in 256 1024; do for pad in 0 12000 24000 40000; do
  timeout -k 10 200 python bench.py --backend rwkv7 --streams $S --lanes 1 --lane-streams $S --sample-tokens 512 --steps 1 --warmup 0 --cpu-sample 0 --debug-set rw_state_pad=$pad > $O/rw_$S_$pad.json 2> $O/rw_$S_$pad.err
  python - <<PY
import json
d=json.loads(open("$O/rw_$S_$pad.json").read().strip().splitlines()[-1])
r=d["roofline"]; print("streams", $S, "pad", $pad, "state avg_us", r["avg_launch_us"], "GB/s", r["achieved"], "| decompress", d["decompress_tokens_per_s_rank0"], "| launches", r["launches"], "alg", r["algorithmic_per_launch"])
PY
done; done
exit 0
