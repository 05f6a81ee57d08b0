#!/bin/bash
# round-4 GPU call 3: phase stamps inside the gemm_mid kernels of the wide step (make mid_dbg): round 3's kernels (mid_cpw = 0) against the new ones
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4c3; mkdir -p $O
cd $R
export CANDLEZIP_AMD_LIB=$R/candlezip_amd/lib/libcandlezip_hip_middbg.so
timeout -k 10 200 python3 tools/mid_phases.py 512 767 20 0 > $O/phases_cpw0.json 2> $O/err0.txt || { tail -5 $O/err0.txt; exit 1; }
timeout -k 10 200 python3 tools/mid_phases.py 512 767 20 -1 > $O/phases_new.json 2> $O/err1.txt || { tail -5 $O/err1.txt; exit 1; }
python3 - <<'PY'
import json,os
O=os.environ.get("GRAFT_REPO_ROOT")+"/gpurun_out/r4c3"
a=json.load(open(O+"/phases_cpw0.json")); b=json.load(open(O+"/phases_new.json"))
print("ms_per_step", a["ms_per_step"], b["ms_per_step"])
for k in a["kernels"]:
    print(k); print("  old", a["kernels"][k]); print("  new", b["kernels"].get(k))
PY
