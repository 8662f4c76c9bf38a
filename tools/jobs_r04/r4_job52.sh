#!/bin/bash
# round-4 GPU job 52: the fuzz test with its row-wise check: committed cases, then campaigns
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/j52
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "fuzz" 2>&1 | tail -4
for seed in 601 602; do
  FUZZ_MANY=1 timeout -k 10 400 python tools/fuzz_campaign.py $seed 260 2>&1 | grep -v amdgpu.ids | tail -8 | tee -a gpurun_out/j52/fuzz_campaign.log
done
timeout -k 10 300 python tools/fuzz_campaign.py 611 300 2>&1 | grep -v amdgpu.ids | tail -8 | tee -a gpurun_out/j52/fuzz_campaign.log
