#!/bin/bash
# round-4 GPU job 54: final measurement pass of the round (m7) behind a parity check of the last changes
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out/m7
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/m7/tests.log 2>&1; echo "tests rc=$?"; tail -2 gpurun_out/m7/tests.log
bash tools/measure_r4.sh gpurun_out/m7 m7 2>&1 | tail -30
export FILTERNAN_FRACS=0.0,0.0001,0.001,0.01,0.05
for m in 0 -1; do
  FILTERNAN_IMPUTE=$m timeout -k 10 500 python tools/filternan.py Matern32x2 Matern52x2 Matern32x4 Matern52x3 Matern52x4 > gpurun_out/m7/filternan_$m.log 2>&1; echo "filternan $m done"
done
timeout -k 10 200 python tools/gemm_probe.py --dtype f32 2>&1 | grep -v amdgpu.ids | tee gpurun_out/m7/gemm_probe.log
timeout -k 10 200 python tools/gemm_probe.py --dtype f32 --T 4096 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/m7/gemm_probe.log
timeout -k 10 200 python tools/gemm_probe.py --dtype f64 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/m7/gemm_probe.log
echo "== done"
for seed in 401 701; do
  FUZZ_MANY=1 timeout -k 10 400 python tools/fuzz_campaign.py $seed 260 2>&1 | grep -v amdgpu.ids | tail -6 | tee -a gpurun_out/m7/fuzz_campaign.log
done
