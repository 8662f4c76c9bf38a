"""Wider random campaign on the GPU box: drives the bodies of the committed fuzz tests (tests/test_gpu_parity.py) with fresh seeds,
longer streams and every segment boundary.  usage: python tools/fuzz_campaign.py [seed] [cases]  (round 3, final library: 2100 cases over 3 seeds, 0 failures; round 4: see profiles/r04/fuzz_campaign.log)"""
import sys, os, numpy as np, torch, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import test_gpu_parity as tp
torch.cuda.set_device(0)
from multioutputihgp_amd import MOIHGP, load_library, streams
from oracle import cref
env = dict(MOIHGP=MOIHGP, streams=streams, cref=cref, lib=load_library())
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 12345)
nfail = 0; n = 0
Ts = [1, 2, 3, 15, 16, 17, 31, 32, 33, 255, 256, 257, 511, 512, 513, 1023, 1024, 1025, 1535, 1536, 2047, 2048, 2049, 3071, 3072, 3073, 4095, 4096, 4097, 5000, 8192, 8193, 10000, 10240, 10241, 12288, 12289, 14336, 14337, 16384, 16385]
kerns_f = ["Matern32", "Matern52", "Matern52x2", "Matern32x2", "Matern52x3", "Matern52x4", "Matern32x4"]
for i in range(int(sys.argv[2]) if len(sys.argv) > 2 else 150):
    T = int(rng.choice(Ts + [int(rng.integers(1, 6000))]))
    L = int(rng.integers(1, 12)) if rng.random() < (0.3 if os.environ.get("FUZZ_MANY") == "1" else 0.8) else int(rng.integers(1024, 1100))   # FUZZ_MANY=1: mostly the many-latent paths
    nanf = float(rng.choice([0.0, 0.0, 0.0, 0.001, 0.01, 0.3]))          # 0.001: sparse gaps (broken links / one walked segment)
    dt_ = "f64" if rng.random() < 0.5 else "f32"
    seed = int(rng.integers(0, 2 ** 31))
    kern = kerns_f[i % len(kerns_f)]
    try:
        tp.test_filter_fuzz_vs_oracle(env, kern, L, T, nanf, dt_, seed); n += 1
    except Exception as e:
        nfail += 1; print("FILTER FAIL", kern, L, T, nanf, dt_, seed, repr(e)[:200])
    if "x" not in kern:
        Lg = min(L, 40)
        try:
            tp.test_gradstream_fuzz_vs_oracle(env, kern, Lg, T, nanf if nanf < 0.3 else 0.02, dt_, seed); n += 1
        except Exception as e:
            nfail += 1; print("GRAD FAIL", kern, Lg, T, nanf, dt_, seed, repr(e)[:200])
    else:
        Lg = min(L, 6)
        try:
            tp.test_stacked_gradstream_vs_oracle(env, kern, torch.float64 if dt_ == "f64" else torch.float32, Lg, T, nanf if nanf < 0.3 else 0.02); n += 1
        except Exception as e:
            nfail += 1; print("STACKED GRAD FAIL", kern, Lg, T, nanf, dt_, repr(e)[:200])
print("campaign done:", n, "cases,", nfail, "failures")
