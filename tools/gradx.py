#!/usr/bin/env python3
"""Gradient sweep (A2 + A5) of the stacked models over long streams: wall time per sweep.
usage: python tools/gradx.py [L] [T] [kernel ...]      (MOIHGP_GRADX_SCAN_FROM=99999999 forces the tick-by-tick kernel; GRADX_NAN=f: fraction of missing ticks)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import synth_params, synth_stream, SEED
from multioutputihgp_amd.streams import LatentBank
L = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
T = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
kernels = sys.argv[3:] or ["Matern52x4", "Matern52x2"]
dev = torch.device("cuda", 0)
for kern in kernels:
    bank = LatentBank(0.1, synth_params(L, 0, np.random.default_rng(SEED), kernel=kern), kernel=kern)
    for dtype in (torch.float64, torch.float32):
        Ty = synth_stream(L, 0, T, dtype, dev, SEED + 1)
        nan = float(os.environ.get("GRADX_NAN", "0"))                 # fraction of missing ticks
        if nan > 0:
            Ty[torch.rand(Ty.shape, device=dev) < nan] = float("nan")
        for want in (False, True):
            for _ in range(2):
                r = bank.grad(Ty, T=T, want_yhat=want)
            torch.cuda.synchronize()
            n = 3
            t0 = time.perf_counter()
            for _ in range(n):
                r = bank.grad(Ty, T=T, want_yhat=want)
            torch.cuda.synchronize()
            ms = 1e3 * (time.perf_counter() - t0) / n
            print(f"{kern} d={bank.d} P={bank.P} {str(dtype)[6:]} L={L} T={T} means={'yes' if want else 'no '}: {ms:9.3f} ms per sweep"
                  f"  ({L * T / ms / 1e6:.1f} G ticks/s)  nll sum {float(r['nll'].sum()):.6e}" + (f"  missing {nan}" if nan else ""), flush=True)
