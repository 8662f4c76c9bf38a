#!/usr/bin/env python3
"""d = 3 filter at a few hundred latents: automatic time split against no split (MOIHGP_FILTER_SPLIT=1), kernel time by HIP events.
Run once per setting (the hook is read once per process):  python tools/micro/split_threshold.py   and   MOIHGP_FILTER_SPLIT=1 python ..."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from bench import synth_params, synth_stream, SEED
from multioutputihgp_amd.streams import LatentBank
dev = torch.device("cuda", 0)
T = 10000
for dtype in (torch.float64, torch.float32):
    for L in (256, 384, 512, 600, 682, 683, 768, 900, 1023, 1024):
        bank = LatentBank(0.1, synth_params(L, 0, np.random.default_rng(SEED)), kernel="Matern52ss")
        Ty = synth_stream(L, 0, T, dtype, dev, SEED + 1)
        yh = torch.empty_like(Ty); n = torch.empty((L,), dtype=torch.float64, device=dev)
        x = torch.zeros((L, 3), dtype=dtype, device=dev); xz = torch.zeros_like(x)
        for _ in range(3): bank.filter(Ty, T=T, x=x, x_start=xz, yhat=yh, nll=n)
        bank.profile_enable(30)
        for _ in range(30): bank.filter(Ty, T=T, x=x, x_start=xz, yhat=yh, nll=n)
        torch.cuda.synchronize()
        ms = float(np.mean(bank.profile_read()))
        print(f"{str(dtype)[6:]} L={L}: {ms*1e3:7.2f} us", flush=True)
