#!/bin/bash
# round-4 GPU job 35: GEMM with LDS write swizzle + fp32 double buffering: parity and rate
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/j35
O=$PWD/gpurun_out/j35
echo "== gemm probe"
timeout -k 10 300 python tools/gemm_probe.py --dtype f32 2>&1 | grep -v amdgpu.ids | tee $O/gemm_f32.log
timeout -k 10 300 python tools/gemm_probe.py --dtype f32 --T 4096 2>&1 | grep -v amdgpu.ids | tee -a $O/gemm_f32.log
timeout -k 10 300 python tools/gemm_probe.py --dtype f64 2>&1 | grep -v amdgpu.ids | tee $O/gemm_f64.log
echo "== tests"
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -x -q -k "project or polar or update or gemm or c3_learning or sharded or abi or window" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -4 $O/tests.log
