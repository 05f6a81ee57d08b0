#!/bin/bash
# round-4 GPU call 28: the rwkv7 bench line as a two-stage pipeline (compress of the next batch beside the decompress of the current one) against one after the other
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4c28; mkdir -p $O
cd $R
for ov in 1 0; do
  timeout -k 10 400 python3 bench.py --backend rwkv7 --overlap $ov --cpu-sample 0 --single-stream-tokens 0 > $O/rwkv7_overlap$ov.json 2> $O/rwkv7_overlap$ov.err; rc=$?
  tail -2 $O/rwkv7_overlap$ov.err | cut -c1-160; [ $rc -ne 0 ] && exit 1
  python3 -c "
import json;d=json.loads(open('$O/rwkv7_overlap$ov.json').read().strip().splitlines()[-1]);print('overlap $ov:', d['value'], d['compress_tokens_per_s_rank0'], d['decompress_tokens_per_s_rank0'], d['steps'], d['ms_per_step'])"
done
