#!/usr/bin/env python3
"""Filter + NLL sweep with and without missing ticks: wall time per sweep (all passes; median of three runs of ten sweeps).  usage: python tools/filternan.py [kernel ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import synth_params, synth_stream, SEED
from multioutputihgp_amd.streams import LatentBank
L, T = 4096, 10000
dev = torch.device("cuda", 0)
IMPUTE = os.environ.get("FILTERNAN_IMPUTE")           # -1 / 0 / 1: option "filter_impute" of the stacked models (unset: the library's default)
for kern in (sys.argv[1:] or ["Matern52ss", "Matern52x2", "Matern52x4"]):
    bank = LatentBank(0.1, synth_params(L, 0, np.random.default_rng(SEED), kern), kernel=kern)
    if IMPUTE is not None and bank.stacked:
        bank.set_option("filter_impute", int(IMPUTE))
    for dtype in ((torch.float64, torch.float32) if "FILTERNAN_DTYPE" not in os.environ else (getattr(torch, os.environ["FILTERNAN_DTYPE"]),)):
        for nan in ((0.0, 0.0001, 0.01, 0.05) if "FILTERNAN_FRACS" not in os.environ else [float(v) for v in os.environ["FILTERNAN_FRACS"].split(",")]):
            Ty = synth_stream(L, 0, T, dtype, dev, SEED + 1)
            if nan > 0:
                Ty[torch.rand(Ty.shape, device=dev) < nan] = float("nan")
            yh = torch.empty_like(Ty); n = torch.empty((L,), dtype=torch.float64, device=dev)
            x = torch.zeros((L, bank.d), dtype=dtype, device=dev); xz = torch.zeros_like(x)
            for _ in range(2):
                bank.filter(Ty, T=T, x=x, x_start=xz, yhat=yh, nll=n)
            torch.cuda.synchronize()
            reps = []
            for _ in range(3):                                     # median of three runs of ten sweeps
                t0 = time.perf_counter()
                for _ in range(10):
                    bank.filter(Ty, T=T, x=x, x_start=xz, yhat=yh, nll=n)
                torch.cuda.synchronize()
                reps.append((time.perf_counter() - t0) / 10 * 1e3)
            ms = sorted(reps)[1]
            print(f"{kern} d={bank.d} {str(dtype)[6:]} nan={nan}: {ms * 1e3:9.1f} us per sweep", flush=True)
