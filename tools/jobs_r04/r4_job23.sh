#!/bin/bash
# round-4 GPU job 23: imputation against the chunk-map path at d = 4 and d = 6; full parity file
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out/j23
O=$PWD/gpurun_out/j23
export FILTERNAN_FRACS=0.0,0.0001,0.001,0.01,0.05
for m in 0 1; do
  echo "-- filter_impute=$m" | tee -a $O/progress.log
  FILTERNAN_IMPUTE=$m timeout -k 10 300 python tools/filternan.py Matern32x2 Matern32x3 Matern52x2 Matern52x4 > $O/filternan_$m.log 2>&1; cat $O/filternan_$m.log
done
echo "== tests" | tee -a $O/progress.log
timeout -k 10 1100 python -m pytest tests/test_gpu_parity.py -x -q > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/progress.log; tail -6 $O/tests.log
echo "== done" | tee -a $O/progress.log
