#!/bin/bash
# round-3 GPU call 21: does an attention-only lane hide a chain-only lane (and the reverse)?  Heterogeneous lanes, room beside the attention blocks (pad=1: one block per CU), stream priorities
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3c21; mkdir -p $O
cd $R
W="wm=256,wmode=3"
timeout -k 10 600 python tools/bench_decode_step.py --iters 40 --max-streams 512 --sym 0 \
  "attn_1l:B=512,L=1,av=2,$W" "chain_1l:B=512,L=1,av=1,$W" "attn_2l:B=512,L=2,av=2,$W" "chain_2l:B=512,L=2,av=1,$W" \
  "attn+chain:B=512,L=2,av=2,av1=1,$W" "attn+chain_pad:B=512,L=2,av=2,av1=1,pad=1,$W" "attn+chain_hi:B=512,L=2,av=2,av1=1,pr1=-1,$W" "attn_lo+chain_hi:B=512,L=2,av=2,av1=1,pr0=1,pr1=-1,$W" \
  "attn+chain_pad_hi:B=512,L=2,av=2,av1=1,pad=1,pr1=-1,$W" "both_2l:B=512,L=2,$W" "both_2l_pad:B=512,L=2,pad=1,$W" > $O/steps.jsonl 2> $O/steps.err; rc=$?; cut -c1-330 $O/steps.jsonl; tail -3 $O/steps.err
exit $rc
