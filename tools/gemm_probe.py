#!/usr/bin/env python3
"""project_stream / unproject_stream at M = L = 4096 over T ticks, fp32 or fp64: wall time per call (torch events) and the MFMA fraction.
usage: python tools/gemm_probe.py [--dtype f32|f64] [--T 10000] [--reps 10]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from multioutputihgp_amd import MOIHGP
from multioutputihgp_amd.streams import project_stream, unproject_stream
ap = argparse.ArgumentParser(); ap.add_argument("--dtype", default="f32"); ap.add_argument("--T", type=int, default=10000); ap.add_argument("--reps", type=int, default=10)
ap.add_argument("--M", type=int, default=4096)
a = ap.parse_args()
dt = torch.float32 if a.dtype == "f32" else torch.float64
M = L = a.M
gp = MOIHGP(0.1, M, L, kernel="Matern52ss")
Y = torch.randn((a.T, M), device="cuda", dtype=dt)
for _ in range(2):
    Ty = project_stream(gp, Y); Yh = unproject_stream(gp, Ty, a.T)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
tp = tu = 0.0
for _ in range(a.reps):
    ev[0].record(); Ty = project_stream(gp, Y); ev[1].record(); Yh = unproject_stream(gp, Ty, a.T); ev[2].record(); torch.cuda.synchronize()
    tp += ev[0].elapsed_time(ev[1]); tu += ev[1].elapsed_time(ev[2])
tp /= a.reps; tu /= a.reps
fl = 2.0 * M * L * a.T; peak = 157.3 if a.dtype == "f32" else 78.6
print(f"{a.dtype} M=L={M} T={a.T}: project {tp:.3f} ms = {fl/tp/1e9:.1f} TF ({fl/tp/1e9/peak*100:.1f} %), unproject {tu:.3f} ms = {fl/tu/1e9:.1f} TF ({fl/tu/1e9/peak*100:.1f} %)")
