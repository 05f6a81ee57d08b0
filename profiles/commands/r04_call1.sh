#!/bin/bash
# round-4 GPU call 1: the GPU suite at the round's first commit (ADVICE fixes), then ONE rocprofv3 --kernel-trace run of the graph-replay
# pipeline WITH the process map dumped after the pipeline fill: the run is expected to die inside the profiler (profiles/r04_profiler_crash/);
# the map says what lies at the fault address.  Not repeated.
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4c1; mkdir -p $O
cd $R && timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?
tail -3 $O/pytest.log
[ $rc -ne 0 ] && exit 1
cd /tmp && export TMPDIR=/tmp
CZ_DUMP_MAPS=$O/maps.txt timeout -k 10 240 rocprofv3 --kernel-trace --output-format csv -d $O/kt -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-sample 0 --profile-tokens 0 --no-events > $O/bench.json 2> $O/bench.err
echo "profiled pipeline rc=$?"; grep -n "SIGSEGV\|PC:" $O/bench.err | cut -c1-200
rm -rf $O/kt
exit 0
