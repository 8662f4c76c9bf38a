#!/bin/bash
# round-4 GPU job 28: fp32 banks -- latents with unusable fp32 tables swept in fp64 on the side: test, timing
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/j28
O=$PWD/gpurun_out/j28
echo "== tests" | tee $O/progress.log
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -k "fp32_bank_sweeps or stacked or unstable" > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/progress.log; tail -12 $O/tests.log
echo "== timing" | tee -a $O/progress.log
export FILTERNAN_FRACS=0.0,0.01
FILTERNAN_DTYPE=float32 timeout -k 10 300 python tools/filternan.py Matern32x2 Matern32x3 Matern32x4 Matern52x2 Matern52x4 > $O/filternan.log 2>&1; grep -v amdgpu.ids $O/filternan.log
echo "== done" | tee -a $O/progress.log
