#!/bin/bash
# round-4 GPU job 47: random parity campaign on the final library, weighted towards the many-latent paths (imputation, side sweep)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/j47
for seed in 401 402; do
  FUZZ_MANY=1 timeout -k 10 500 python tools/fuzz_campaign.py $seed 260 2>&1 | grep -v amdgpu.ids | tail -12 | tee -a gpurun_out/j47/fuzz_campaign.log
done
timeout -k 10 300 python tools/fuzz_campaign.py 403 300 2>&1 | grep -v amdgpu.ids | tail -6 | tee -a gpurun_out/j47/fuzz_campaign.log
