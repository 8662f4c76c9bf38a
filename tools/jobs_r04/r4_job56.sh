cd "$GRAFT_REPO_ROOT"
for v in bk16 bk64; do echo "== $v"; MOIHGP_LIB=$PWD/multioutputihgp_amd/lib/libmoihgp_$v.so timeout -k 10 200 python tools/gemm_probe.py --dtype f32 2>&1 | grep -v amdgpu.ids; MOIHGP_LIB=$PWD/multioutputihgp_amd/lib/libmoihgp_$v.so timeout -k 10 200 python tools/gemm_probe.py --dtype f64 2>&1 | grep -v amdgpu.ids; done
