#!/usr/bin/env python3
"""Singular values of the matrix whose Gram matrix MOIHGP_POLAR_DUMP wrote (raw doubles, L x L): sqrt of eigvalsh, summary + .npy.
usage: python tools/polar_spectrum.py gram.bin [out.npy]"""
import sys
import numpy as np
g = np.fromfile(sys.argv[1], dtype=np.float64)
L = int(round(np.sqrt(g.size)))
w = np.linalg.eigvalsh(g.reshape(L, L))
sv = np.sqrt(np.clip(w, 0, None))
print(f"L={L} sigma min {sv.min():.6f} max {sv.max():.6f} median {np.median(sv):.6f}")
print("quantiles 0,1,5,25,50,75,95,99,100 %:", np.round(np.percentile(sv, [0, 1, 5, 25, 50, 75, 95, 99, 100]), 5).tolist())
print("largest 40:", np.round(sv[-40:], 4).tolist())
print("smallest 10:", np.round(sv[:10], 4).tolist())
print("count sigma > 1.05:", int((sv > 1.05).sum()), " > 1.5:", int((sv > 1.5).sum()), " < 0.95:", int((sv < 0.95).sum()))
if len(sys.argv) > 2:
    np.save(sys.argv[2], sv)
