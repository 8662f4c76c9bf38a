#!/bin/bash
# round-4 GPU job 32: second measurement pass of the round (final kernels) + the missing-data tables again (median of three)
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out/m3
bash tools/measure_r4.sh gpurun_out/m3 m3 2>&1 | tail -40
export FILTERNAN_FRACS=0.0,0.0001,0.001,0.01,0.05
for m in 0 -1; do
  FILTERNAN_IMPUTE=$m timeout -k 10 500 python tools/filternan.py Matern32x2 Matern52x2 Matern32x4 Matern52x3 Matern52x4 > gpurun_out/m3/filternan_$m.log 2>&1; echo "filternan $m done"
done
echo "== done"
