#!/usr/bin/env python3
"""Why the imputation sweep does or does not take a bank's latents: one sweep with MOIHGP_GAP_TRACE=1.  usage: python tools/gapdbg.py [kernel ...]"""
import os, sys
os.environ["MOIHGP_GAP_TRACE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import synth_params, synth_stream, SEED
from multioutputihgp_amd.streams import LatentBank
L, T = 4096, 10000
dev = torch.device("cuda", 0)
for kern in (sys.argv[1:] or ["Matern32x2", "Matern32x3", "Matern52x2", "Matern52x4"]):
    bank = LatentBank(0.1, synth_params(L, 0, np.random.default_rng(SEED), kern), kernel=kern)
    bank.set_option("filter_impute", 1)
    for dtype in (torch.float64, torch.float32):
        Ty = synth_stream(L, 0, T, dtype, dev, SEED + 1)
        Ty[torch.rand(Ty.shape, device=dev) < 0.01] = float("nan")
        print(kern, dtype, flush=True)
        bank.filter(Ty, T=T)
        torch.cuda.synchronize()
