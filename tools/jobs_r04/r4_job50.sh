#!/bin/bash
# round-4 GPU job 50: after the growth bound: the new test, the imputation tests, a many-latent fuzz campaign
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/j50
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "imputation or unstable or fp32_bank or stacked_filter_gaps or stacked_missing" 2>&1 | tail -4
for seed in 401 404 405; do
  FUZZ_MANY=1 timeout -k 10 500 python tools/fuzz_campaign.py $seed 260 2>&1 | grep -v amdgpu.ids | tail -6 | tee -a gpurun_out/j50/fuzz_campaign.log
done
