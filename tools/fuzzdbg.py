#!/usr/bin/env python3
"""One case of the filter fuzz test (tests/test_gpu_parity.py::test_filter_fuzz_vs_oracle) with the worst latents named: which of them differ from the oracle,
by how much, their rho(AKHA), and which path took them (MOIHGP_GAP_TRACE).  usage: python tools/fuzzdbg.py kernel L T nanf f32|f64 seed"""
import os, sys
os.environ["MOIHGP_GAP_TRACE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import test_gpu_parity as tp
from multioutputihgp_amd import streams
from oracle import cref
kern, L, T, nanf, dt_, seed = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4]), sys.argv[5], int(sys.argv[6])
rng = np.random.default_rng(seed)
dtype = torch.float64 if dt_ == "f64" else torch.float32
prm = tp.synth_params_stacked(L, int(kern[-1]), rng) if "x" in kern else tp.synth_params(L, rng)
bank = streams.LatentBank(0.1, prm, kernel=kern if "x" in kern else tp.KMAP[kern])
if len(sys.argv) > 7:
    bank.set_option("filter_impute", int(sys.argv[7]))
igps = cref.ihgp_array(kern, 0.1, prm)
x0 = 0.3 * rng.standard_normal((L, bank.d))
Ty = tp.synth(L, T, rng, nan_frac=nanf)
o = cref.filter_stream(igps, Ty, x0=x0, nthreads=8)
yhat, xT, nll = bank.filter(tp.to_dev(Ty, dtype), T=T, x=torch.from_numpy(x0).to(dtype).cuda())
torch.cuda.synchronize()
yo = o["yhat"]; yg = yhat[:, :T].cpu().numpy().astype(np.float64)
big = np.nan_to_num(np.abs(yo), nan=0.0, posinf=np.inf).max(axis=1)
tame = big < (1e100 if dtype == torch.float64 else 1e20)
scale = np.abs(yo[tame]).max()
err = np.abs(yg - yo).max(axis=1)
rho = np.array([max(abs(np.linalg.eigvals(g.mat("AKHA")))) for g in igps])
print("global scale", scale, " tame", tame.sum(), "of", L)
for l in np.argsort(-np.where(tame, err, 0))[:8]:
    t = int(np.argmax(np.abs(yg[l] - yo[l])))
    print(f"latent {l}: max|yhat| {big[l]:.3e}  err {err[l]:.3e} (rel to own max {err[l] / max(big[l], 1e-300):.2e}, to global {err[l] / scale:.2e}) at tick {t}  rho {rho[l]:.5f}  gaps {int(np.isnan(Ty[l]).sum())}  nll gpu {nll[l].item():.6e} oracle {o['nll_per_latent'][l]:.6e}")
xe = np.abs(xT.cpu().numpy() - o["x"]).max(axis=1)
print("state: worst tame", np.where(tame, xe, 0).max() / max(np.abs(o["x"][tame]).max(), 1e-300))
