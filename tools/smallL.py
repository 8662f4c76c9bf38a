#!/usr/bin/env python3
"""Few-latents filter paths side by side: automatic time split (slices of uneven length, balanced over the SIMDs) vs forced equal slices
(MOIHGP_FILTER_SPLIT=7 / 8) vs no split (MOIHGP_FILTER_SPLIT=1): agreement of the results and kernel-exact durations (dispatch-attached HIP events).
usage: python tools/smallL.py [--L 256] [--T 10000] [--dtype f64] [--kernel Matern52ss] [--nan 0.0]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from bench import synth_params, synth_stream, SEED
from multioutputihgp_amd.streams import LatentBank

ap = argparse.ArgumentParser()
ap.add_argument("--L", type=int, default=256); ap.add_argument("--T", type=int, default=10000); ap.add_argument("--dtype", default="f64")
ap.add_argument("--kernel", default="Matern52ss"); ap.add_argument("--nan", type=float, default=0.0)
ap.add_argument("--reps", type=int, default=30); ap.add_argument("--mode", default="fn")
a = ap.parse_args()
dtype = torch.float32 if a.dtype == "f32" else torch.float64
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
prm = synth_params(a.L, 0, np.random.default_rng(SEED), a.kernel)
bank = LatentBank(0.1, prm, kernel=a.kernel)
Ty = synth_stream(a.L, 0, a.T, dtype, dev, SEED + 1)
if a.nan > 0:
    Ty[torch.rand(Ty.shape, device=dev) < a.nan] = float("nan")
x0 = 0.1 * torch.randn((a.L, bank.d), dtype=dtype, device=dev)
es = 4 if dtype == torch.float32 else 8
nbytes = (("f" in a.mode) + 1) * es * a.L * a.T


def run(env):
    for k in ("MOIHGP_FILTER_SPLIT",):
        os.environ.pop(k, None)
    os.environ.update(env)
    yhat = torch.empty_like(Ty); nll = torch.empty((a.L,), dtype=torch.float64, device=dev); x = x0.clone()
    tot = torch.zeros(1, dtype=torch.float64, device=dev)
    for _ in range(3):
        x.copy_(x0); bank.filter(Ty, T=a.T, x=x, yhat=yhat, nll=nll, want_yhat="f" in a.mode, want_nll="n" in a.mode, nll_total=tot)
    bank.profile_enable(a.reps)
    torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    for _ in range(a.reps):
        bank.filter(Ty, T=a.T, x=x, x_start=x0, yhat=yhat, nll=nll, want_yhat="f" in a.mode, want_nll="n" in a.mode, nll_total=tot)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / a.reps * 1e6
    t = np.array(bank.profile_read()) * 1e3
    return dict(yhat=yhat[:, :a.T].clone(), nll=nll.clone(), x=x.clone(), t=t, wall=wall, tot=tot.item())


cases = [("auto", {}), ("equal7", {"MOIHGP_FILTER_SPLIT": "7"}), ("equal8", {"MOIHGP_FILTER_SPLIT": "8"}), ("whole", {"MOIHGP_FILTER_SPLIT": "1"})]
res = {n: run(e) for n, e in cases}
ref = res["whole"]
print(f"L={a.L} T={a.T} {a.dtype} {a.kernel} nan={a.nan} mode={a.mode}: {nbytes/1e6:.1f} MB per launch")
for n, r in res.items():
    def rel(u, v):
        m = torch.isfinite(v)
        return float((u[m] - v[m]).abs().max() / v[m].abs().max())
    e = (rel(r["yhat"], ref["yhat"]) if "f" in a.mode else 0.0, rel(r["nll"], ref["nll"]) if "n" in a.mode else 0.0, rel(r["x"], ref["x"]))
    t = r["t"]
    print(f"{n:12s} kernel min {t.min():6.1f} med {np.median(t):6.1f} us  {nbytes/np.median(t)/1e6:5.2f} TB/s = {nbytes/np.median(t)/1e6/8*100:4.1f}%  wall/pass {r['wall']:6.1f} us"
          f"   vs whole: yhat {e[0]:.1e} nll {e[1]:.1e} x {e[2]:.1e}  total {r['tot']:.10e}")
