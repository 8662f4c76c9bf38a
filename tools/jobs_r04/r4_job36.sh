#!/bin/bash
# round-4 GPU job 36: GEMM variants (double buffer x write swizzle), fp32
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/j36
for v in 00 01 10 11; do
  echo "== DB SWZ = $v"
  MOIHGP_LIB=$PWD/multioutputihgp_amd/lib/libmoihgp_g$v.so timeout -k 10 300 python tools/gemm_probe.py --dtype f32 2>&1 | grep -v amdgpu.ids
  MOIHGP_LIB=$PWD/multioutputihgp_amd/lib/libmoihgp_g$v.so timeout -k 10 300 python tools/gemm_probe.py --dtype f64 2>&1 | grep -v amdgpu.ids
done
