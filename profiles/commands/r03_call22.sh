#!/bin/bash
# round-3 GPU call 22: cost of an all-to-all hand-off inside ONE launch (persistent blocks, agent-scope counter), all XCDs and one XCD
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3c22; mkdir -p $O
cd $R
timeout -k 10 300 python tools/probe_persist.py > $O/persist.jsonl 2> $O/persist.err; rc=$?; cat $O/persist.jsonl; tail -3 $O/persist.err
exit $rc
