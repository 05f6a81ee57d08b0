#!/bin/bash
# round-2 GPU call 17: encode alone; SQ counters of the prefill attention and the big-M GEMM
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2c17; mkdir -p $O
cd $R
timeout -k 10 200 python tools/encode_only.py 192 2048 2 | tee $O/encode_only.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -oE "\b(SQ_[A-Z_0-9]+|GRBM_[A-Z_]+|TCC_HIT_sum|TCC_MISS_sum|MfmaUtil|VALUBusy|LDSBankConflict)\b" | sort -u > $O/counters.txt; wc -l $O/counters.txt
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_WAVES"; do
  tag=$(echo $set | cut -c1-12 | tr ' ' '_')
  timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --kernel-include-regex "attention_v2_kernel|gemm_bf16w_kernel" --output-format csv -d /tmp/p_$tag -o pmc -- python3 $R/tools/encode_only.py 96 2048 1 > $O/pmc_$tag.out 2> $O/pmc_$tag.err; rc=$?
  echo "pmc [$set] rc=$rc"; tail -2 $O/pmc_$tag.err | cut -c1-300
  [ $rc -eq 0 ] && python3 $R/tools/prof_summary.py /tmp/p_$tag $O/sq_$tag
  [ $rc -ge 124 ] && exit 1
done
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('/root/repo/gpurun_out/r2c17/sq_*_pmc.json')):
    d=json.load(open(f))
    for k,v in d.items():
        print(k[:44], {c: round(x['mean_per_dispatch']) for c,x in v.items()})
PY
exit 0
