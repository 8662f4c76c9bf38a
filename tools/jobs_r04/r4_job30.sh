#!/bin/bash
# round-4 GPU job 30: imputation against the chunk-map second pass at d = 4 and d = 6 (should "automatic" cover them?)
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/j30
O=$PWD/gpurun_out/j30
export FILTERNAN_FRACS=0.0,0.0001,0.001,0.01,0.05
for m in 0 1; do
  echo "-- filter_impute=$m"
  FILTERNAN_IMPUTE=$m timeout -k 10 300 python tools/filternan.py Matern32x2 Matern32x3 Matern52x2 > $O/filternan_$m.log 2>&1; grep -v amdgpu.ids $O/filternan_$m.log
done
