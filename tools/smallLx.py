#!/usr/bin/env python3
"""Stacked filter at few latents: kernel time (HIP events) against the number of latents.  usage: python tools/smallLx.py [kernel] [T]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import synth_params, synth_stream, SEED
from multioutputihgp_amd.streams import LatentBank
kern = sys.argv[1] if len(sys.argv) > 1 else "Matern52x2"
T = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
dev = torch.device("cuda", 0)
for L in (64, 128, 192, 224, 227, 256, 288, 384, 512):
    bank = LatentBank(0.1, synth_params(L, 0, np.random.default_rng(SEED), kern), kernel=kern)
    Ty = synth_stream(L, 0, T, torch.float64, dev, SEED + 1)
    yh = torch.empty_like(Ty); n = torch.empty((L,), dtype=torch.float64, device=dev)
    x = torch.zeros((L, bank.d), dtype=torch.float64, device=dev); xz = torch.zeros_like(x)
    for _ in range(3):
        bank.filter(Ty, T=T, x=x, x_start=xz, yhat=yh, nll=n)
    bank.profile_enable(30)
    for _ in range(30):
        bank.filter(Ty, T=T, x=x, x_start=xz, yhat=yh, nll=n)
    torch.cuda.synchronize()
    ms = float(np.mean(bank.profile_read()))
    print(f"{kern} L={L} T={T}: kernel {ms * 1e3:7.2f} us   {16 * L * T / ms / 1e6:8.1f} GB/s = {16 * L * T / ms / 1e6 / 80:.1f} % of 8 TB/s", flush=True)
