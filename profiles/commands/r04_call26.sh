#!/bin/bash
# round-4 GPU call 26: the SmolLM decode CDF with its row streamed through LDS when the launch is a handful of blocks (cdf_dec_stage_rows): parity of the CDF /
# decoder tests and of the narrow-group model tests, the one-row phases (make timers), the single-stream step with and without
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4c26; mkdir -p $O
cd $R && timeout -k 10 800 python3 -m pytest tests/test_gpu_kernels.py tests/test_gpu_round2.py tests/test_gpu_round4.py tests/test_gpu_model.py -m gpu -x -q -k "decoder or decode or gemv or several_streams or trait or smollm2_135m" > $O/pytest.log 2>&1; rc=$?
tail -3 $O/pytest.log
[ $rc -ne 0 ] && exit 1
CANDLEZIP_AMD_LIB=$R/candlezip_amd/lib/libcandlezip_hip_timers.so timeout -k 10 100 python3 tools/cdf_phases.py 49152 0 100 decode > $O/cdf_decode_phases_staged.txt 2> $O/err.txt || { tail -3 $O/err.txt; exit 1; }
cat $O/cdf_decode_phases_staged.txt
timeout -k 10 300 python3 tools/bench_decode_step.py --iters 200 --max-streams 8 b1:B=1 b1b:B=1 b4:B=4 > $O/steps.jsonl 2> $O/steps.err
python3 - <<'PY'
import json,os
for l in open(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r4c26/steps.jsonl"):
    d=json.loads(l); print(d["config"], d["ms_per_step_lane"], d["tokens_per_s_device_time"], d["errors"])
PY
