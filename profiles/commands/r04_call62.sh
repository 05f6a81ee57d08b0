#!/bin/bash
# round-4 GPU call 62: the new GEMM parity test (half-tile pipeline against the one-buffer loop, K = 64 .. 576), then the FETCH_SIZE / WRITE_SIZE passes of the SmolLM profile pass at the final kernels
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4c62; mkdir -p $O
cd $R
timeout -k 10 600 python3 -m pytest tests/test_gpu_round4.py tests/test_gpu_kernels.py -m gpu -x -q > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
cd /tmp && export TMPDIR=/tmp
run() { local name=$1; shift; timeout -k 10 900 rocprofv3 "$@" > $O/$name.out 2> $O/$name.err; local rc=$?; echo "$name rc=$rc"; grep "profile_pass\]" $O/$name.err | tail -1; return $rc; }
run pmc_f_smollm --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_f_smollm -- python3 $R/tools/profile_pass.py smollm --tokens 1024 || exit 1
python3 $R/tools/prof_summary.py $O/pmc_f_smollm $O/r04_pmc1024_FETCH_SIZE --skip-frac 0.5; rm -rf $O/pmc_f_smollm
run pmc_w_smollm --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_w_smollm -- python3 $R/tools/profile_pass.py smollm --tokens 1024 || exit 1
python3 $R/tools/prof_summary.py $O/pmc_w_smollm $O/r04_pmc1024_WRITE_SIZE --skip-frac 0.5; rm -rf $O/pmc_w_smollm
ls $O
