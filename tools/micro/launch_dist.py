#!/usr/bin/env python3
"""Kernel time of consecutive launches of one filter shape (dispatch-attached events): does it drift?  python tools/micro/launch_dist.py [kernel dtype n]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from bench import synth_params, synth_stream, SEED
from multioutputihgp_amd.streams import LatentBank
dev = torch.device("cuda", 0)
kern = sys.argv[1] if len(sys.argv) > 1 else "Matern52x2"
dtype = torch.float32 if (len(sys.argv) > 2 and sys.argv[2] == "f32") else torch.float64
n = int(sys.argv[3]) if len(sys.argv) > 3 else 400
L, T = 4096, 10000
bank = LatentBank(0.1, synth_params(L, 0, np.random.default_rng(SEED), kern), kernel=kern)
Ty = synth_stream(L, 0, T, dtype, dev, SEED + 1)
yhat = torch.empty_like(Ty); nll = torch.empty((L,), dtype=torch.float64, device=dev)
x = torch.zeros((L, bank.d), dtype=dtype, device=dev); xz = torch.zeros_like(x)
for _ in range(5): bank.filter(Ty, T=T, x=x, x_start=xz, yhat=yhat, nll=nll)
bank.profile_enable(n)
for _ in range(n): bank.filter(Ty, T=T, x=x, x_start=xz, yhat=yhat, nll=nll)
t = np.array(bank.profile_read()) * 1e3
print(kern, dtype, "every 8th:", " ".join(f"{v:.0f}" for v in t[::8]))
print("mean %.1f median %.1f min %.1f max %.1f" % (t.mean(), np.median(t), t.min(), t.max()))
