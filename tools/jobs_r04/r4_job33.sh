#!/bin/bash
# round-4 GPU job 33: state form of the gap recursion (long memories, dense gaps): parity, trace, timing
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/j33
O=$PWD/gpurun_out/j33
echo "== tests" | tee $O/progress.log
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -k "imputation or stacked or fp32_bank" > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/progress.log; tail -8 $O/tests.log
echo "== trace" | tee -a $O/progress.log
timeout -k 10 300 python tools/gapdbg.py Matern32x2 Matern32x4 Matern52x4 2>&1 | grep -v amdgpu.ids | tee $O/trace.log
echo "== timing" | tee -a $O/progress.log
export FILTERNAN_FRACS=0.0,0.0001,0.001,0.01,0.05
for m in 1 0; do
  echo "-- filter_impute=$m" | tee -a $O/progress.log
  FILTERNAN_IMPUTE=$m timeout -k 10 500 python tools/filternan.py Matern32x2 Matern32x3 Matern32x4 > $O/filternan_$m.log 2>&1; grep -v amdgpu.ids $O/filternan_$m.log
done
echo "== done" | tee -a $O/progress.log
