#!/usr/bin/env python3
"""Gradient sweep (A2 + A5) with and without missing ticks: wall time per sweep.  usage: python tools/gradnan.py [L] [T]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import synth_params, synth_stream, SEED
from multioutputihgp_amd.streams import LatentBank
L = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
T = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
dev = torch.device("cuda", 0)
bank = LatentBank(0.1, synth_params(L, 0, np.random.default_rng(SEED)), kernel="Matern52ss")
for dtype in (torch.float64, torch.float32):
    for nan in (0.0, 0.0001, 0.01, 0.05):
        Ty = synth_stream(L, 0, T, dtype, dev, SEED + 1)
        if nan > 0:
            Ty[torch.rand(Ty.shape, device=dev) < nan] = float("nan")
        frac = float(torch.isnan(Ty[:, :T]).any(dim=1).double().mean())
        for _ in range(2):
            r = bank.grad(Ty, T=T)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 5
        for _ in range(n):
            r = bank.grad(Ty, T=T)
        torch.cuda.synchronize()
        print(f"{str(dtype)[6:]} L={L} T={T} nan={nan}: {1e3 * (time.perf_counter() - t0) / n:8.3f} ms per sweep  (latents with a missing tick: {100 * frac:.0f} %)", flush=True)
