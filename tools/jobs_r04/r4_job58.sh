#!/bin/bash
# round-4 GPU job 58: more seeds of the random parity campaign on the final library
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/j58
for seed in 801 802 803 804 805; do
  FUZZ_MANY=1 timeout -k 10 400 python tools/fuzz_campaign.py $seed 260 2>&1 | grep -v amdgpu.ids | tail -6 | tee -a gpurun_out/j58/fuzz_campaign.log
done
for seed in 811 812; do
  timeout -k 10 300 python tools/fuzz_campaign.py $seed 300 2>&1 | grep -v amdgpu.ids | tail -6 | tee -a gpurun_out/j58/fuzz_campaign.log
done
