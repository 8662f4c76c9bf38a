#!/usr/bin/env python3
"""Kernel time of the filter + NLL sweep against the stream length at a fixed number of latents: slope = per-tick cost, intercept =
fixed cost of a launch (dispatch of 4096 wavefronts, table loads, first touch of the stream, tail).  usage: python tools/micro/filter_vs_T.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from bench import synth_params, synth_stream, SEED
from multioutputihgp_amd.streams import LatentBank
L = 4096
dev = torch.device("cuda", 0)
bank = LatentBank(0.1, synth_params(L, 0, np.random.default_rng(SEED)), kernel="Matern52ss")
for dtype in (torch.float32, torch.float64):
    res = []
    for T in ([int(t) for t in os.environ["TS"].split(",")] if "TS" in os.environ else (1024, 2048, 4096, 6144, 8192, 10240, 20480)):
        Ty = synth_stream(L, 0, T, dtype, dev, SEED + 1)
        yh = torch.empty_like(Ty); n = torch.empty((L,), dtype=torch.float64, device=dev)
        x = torch.zeros((L, 3), dtype=dtype, device=dev); xz = torch.zeros_like(x)
        for _ in range(3):
            bank.filter(Ty, T=T, x=x, x_start=xz, yhat=yh, nll=n)
        bank.profile_enable(30)
        for _ in range(30):
            bank.filter(Ty, T=T, x=x, x_start=xz, yhat=yh, nll=n)
        torch.cuda.synchronize()
        ms = float(np.mean(bank.profile_read()))
        res.append((T, ms * 1e3))
        print(f"{str(dtype)[6:]} L={L} T={T}: {ms * 1e3:8.2f} us", flush=True)
    Ts = np.array([r[0] for r in res], float); us = np.array([r[1] for r in res])
    if "TS" in os.environ: continue
    a, b = np.polyfit(Ts[:6], us[:6], 1)
    es = 4 if dtype == torch.float32 else 8
    print(f"   fit: {b:.2f} us + {a * 1024:.3f} us per 1024 ticks  ->  asymptotic {2 * es * L / a / 1e6:.2f} TB/s")
