#!/bin/bash
# round-4 GPU job 37: the whole GPU suite, smoke, the missing-data tables (profiles/r04)
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out/j37
O=$PWD/gpurun_out/j37
echo "== tests" | tee $O/progress.log
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/progress.log; tail -4 $O/tests.log
echo "== smoke" | tee -a $O/progress.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
echo "== timing" | tee -a $O/progress.log
export FILTERNAN_FRACS=0.0,0.0001,0.001,0.01,0.05
for m in 0 -1; do
  echo "-- filter_impute=$m" | tee -a $O/progress.log
  FILTERNAN_IMPUTE=$m timeout -k 10 400 python tools/filternan.py Matern32x2 Matern52x2 Matern32x4 Matern52x3 Matern52x4 > $O/filternan_$m.log 2>&1; grep -v amdgpu.ids $O/filternan_$m.log | tail -3
done
echo "== done" | tee -a $O/progress.log
