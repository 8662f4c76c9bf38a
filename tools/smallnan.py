#!/usr/bin/env python3
"""Few latents, streams with missing ticks: kernel time of the filter sweep per fraction of missing ticks, team kernels on / off.
    python tools/smallnan.py [kernel:dtype:L ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import synth_params, synth_stream, SEED
from multioutputihgp_amd.streams import LatentBank
dev = torch.device("cuda", 0)
T = 10000
for sh in (sys.argv[1:] or ["Matern52x2:f64:256", "Matern52x2:f32:256", "Matern32x2:f64:64", "Matern52x4:f32:256"]):
    kern, dt, L = sh.split(":"); L = int(L)
    dtype = torch.float32 if dt == "f32" else torch.float64
    bank = LatentBank(0.1, synth_params(L, 0, np.random.default_rng(SEED), kern), kernel=kern)
    for nan in (0.0, 0.0001, 0.001, 0.01, 0.05):
        Ty = synth_stream(L, 0, T, dtype, dev, SEED + 1)
        if nan > 0:
            Ty[torch.rand(Ty.shape, device=dev) < nan] = float("nan")
        yh = torch.empty_like(Ty); n = torch.empty((L,), dtype=torch.float64, device=dev)
        x = torch.zeros((L, bank.d), dtype=dtype, device=dev); xz = torch.zeros_like(x)
        row = []
        for team in (-1, 0):
            bank.set_option("filter_team", team)
            for _ in range(20): bank.filter(Ty, T=T, x=x, x_start=xz, yhat=yh, nll=n)
            bank.profile_enable(20)
            for _ in range(20): bank.filter(Ty, T=T, x=x, x_start=xz, yhat=yh, nll=n)
            row.append(float(np.median(np.array(bank.profile_read()) * 1e3)))
        print(f"{sh:22s} nan={nan:<7g} team kernels {row[0]:8.1f} us   without {row[1]:8.1f} us", flush=True)
