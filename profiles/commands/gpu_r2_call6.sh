#!/bin/bash
# round-2 GPU call 6: parity of the MFMA sum-of-squares norm path; decode GEMM block shape (fat vs thin) x lanes x hardware queues
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2c6; mkdir -p $O
cd $R
timeout -k 10 400 python -m pytest tests/test_gpu_model.py tests/test_gpu_fuzz.py tests/test_gpu_round2.py -m gpu -x -q > $O/pytest.log 2>&1; rc=$?
tail -4 $O/pytest.log; echo "pytest rc=$rc"
[ $rc -ge 124 ] && exit 1
show() { python - $1 <<'PY'
import json,sys
for l in open(sys.argv[1]):
    d=json.loads(l); print(f"{d['config']:26s} B={d['B']:3d} L={d['lanes']:2d} skip={d['dbg_skip']} fn={d['fuse_norm']} ms/step min={min(d['ms_per_step_lane']):.3f} max={max(d['ms_per_step_lane']):.3f} tok/s={d['tokens_per_s_device_time']} {d['errors'][:1]}")
PY
}
CFGS=""
for gt in 384 96 32; do for lb in "1 128" "3 128" "4 128" "6 64" "8 64"; do set -- $lb; CFGS="$CFGS full_l$1b$2:B=$2,L=$1,gt=$gt gemm_l$1b$2:B=$2,L=$1,av=1,gt=$gt"; done; done
echo "== GPU_MAX_HW_QUEUES default"
timeout -k 10 500 python tools/bench_decode_step.py --iters 100 "nofuse_l1:B=128,L=1,fn=0" "nofuse_l3:B=128,L=3,fn=0" $CFGS > $O/steps_q4.jsonl 2> $O/steps_q4.err; rc=$?
show $O/steps_q4.jsonl; tail -2 $O/steps_q4.err
[ $rc -ge 124 ] && exit 1
echo "== GPU_MAX_HW_QUEUES=8"
GPU_MAX_HW_QUEUES=8 timeout -k 10 500 python tools/bench_decode_step.py --iters 100 $CFGS > $O/steps_q8.jsonl 2> $O/steps_q8.err; rc=$?
show $O/steps_q8.jsonl; tail -2 $O/steps_q8.err
exit 0
