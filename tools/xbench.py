#!/usr/bin/env python3
"""Kernel times (dispatch-attached HIP events) of the filter sweep for a list of shapes -- the quick A/B loop for kernel work.
    python tools/xbench.py [kernel:dtype:L[:T] ...]       default: the stacked rows of the bench line + the d = 3 ones
Prints one line per shape: median / min kernel time over `--n` launches, algorithmic TB/s, share of 8 TB/s, NLL total (a checksum)."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from bench import synth_params, synth_stream, SEED
from multioutputihgp_amd.streams import LatentBank

DEFAULT = ["Matern52x2:f32:4096", "Matern52x2:f64:4096", "Matern52x2:f64:256", "Matern52x4:f64:4096", "Matern52x4:f32:4096", "Matern32x2:f64:4096",
           "Matern52ss:f32:4096", "Matern52ss:f64:4096", "Matern52ss:f64:256"]
ap = argparse.ArgumentParser()
ap.add_argument("shapes", nargs="*", default=DEFAULT)
ap.add_argument("--n", type=int, default=20)
ap.add_argument("--opt", action="append", default=[], help="name=value handle options (moihgp_set_option)")
ap.add_argument("--warm", type=int, default=3)
ap.add_argument("--total", action="store_true", help="also queue the pass's NLL total behind every sweep (as bench.py's passes do)")
a = ap.parse_args()
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
for sh in a.shapes:
    f = sh.split(":")
    kern, dt, L = f[0], f[1], int(f[2]); T = int(f[3]) if len(f) > 3 else 10000
    dtype = torch.float32 if dt == "f32" else torch.float64
    bank = LatentBank(0.1, synth_params(L, 0, np.random.default_rng(SEED), kern), kernel=kern)
    for o in a.opt:
        k, v = o.split("="); bank.set_option(k, int(v))
    Ty = synth_stream(L, 0, T, dtype, dev, SEED + 1)
    yhat = torch.empty_like(Ty); nll = torch.empty((L,), dtype=torch.float64, device=dev)
    x = torch.zeros((L, bank.d), dtype=dtype, device=dev); xz = torch.zeros_like(x)
    tot = torch.zeros(1, dtype=torch.float64, device=dev) if a.total else None
    for _ in range(a.warm):
        bank.filter(Ty, T=T, x=x, x_start=xz, yhat=yhat, nll=nll, nll_total=tot)
    bank.profile_enable(a.n)
    for _ in range(a.n):
        bank.filter(Ty, T=T, x=x, x_start=xz, yhat=yhat, nll=nll, nll_total=tot)
    t = np.array(bank.profile_read()) * 1e3
    es = 4 if dtype == torch.float32 else 8
    b = 2 * es * L * T
    print(f"{sh:26s} median {np.median(t):8.1f} us  min {t.min():8.1f} us   {b / np.median(t) / 1e6:5.2f} TB/s = {b / np.median(t) / 1e6 / 8 * 100:4.1f} %   nll {nll.sum().item():.10e}", flush=True)
    del bank, Ty, yhat
