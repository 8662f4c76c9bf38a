#!/usr/bin/env python3
"""Where one workgroup of the chunk-templated team kernel spends its time: shader-clock stamps of workgroup 0's wavefronts at the stage boundaries.
Needs the tuning build (make -C multioutputihgp_amd/csrc TUNING=1) and MOIHGP_LIB=multioutputihgp_amd/lib/libmoihgp_tuning.so.
    MOIHGP_LIB=... python tools/team_stamps.py [kernel:dtype:L[:T]]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from bench import synth_params, synth_stream, SEED
from multioutputihgp_amd.streams import LatentBank
from multioutputihgp_amd import _lib

STAGES = ["loads issued -> tables in LDS", "stage in", "chunk response", "scan", "carry-in wait", "start states", "replay", "stage out", "nll + stores done"]
lib = ctypes.CDLL(_lib.library_path())            # (same file the bank has open: same symbol instance)
for sh in (sys.argv[1:] or ["Matern52x2:f64:256", "Matern52x2:f64:64", "Matern52x2:f32:256"]):
    f = sh.split(":")
    kern, dt, L = f[0], f[1], int(f[2]); T = int(f[3]) if len(f) > 3 else 10000
    dtype = torch.float32 if dt == "f32" else torch.float64
    dev = torch.device("cuda", 0)
    bank = LatentBank(0.1, synth_params(L, 0, np.random.default_rng(SEED), kern), kernel=kern)
    bank.set_option("filter_team", 1)
    Ty = synth_stream(L, 0, T, dtype, dev, SEED + 1)
    yhat = torch.empty_like(Ty); nll = torch.empty((L,), dtype=torch.float64, device=dev)
    x = torch.zeros((L, bank.d), dtype=dtype, device=dev); xz = torch.zeros_like(x)
    for _ in range(200):
        bank.filter(Ty, T=T, x=x, x_start=xz, yhat=yhat, nll=nll)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 256)()
    assert lib.moihgp_tuning_team_stamps(buf) == 0
    st = np.array(buf, dtype=np.uint64).reshape(16, 16).astype(np.int64)
    ck = next(c for c in (16, 20, 24, 28, 32) if T <= 512 * c); nw = (T + 64 * ck - 1) // (64 * ck)
    t0 = st[:nw, 0].min()
    wall = (st[:nw, 15].max() - st[:nw, 14].min()) * 10.0          # ns (100 MHz)
    cyc = st[:nw, 9].max() - t0
    print(f"{sh}: {nw} wavefronts, first stamp -> last stamp {wall:.0f} ns = {cyc} shader cycles ({cyc / wall:.2f} GHz)")
    print("   wave  start " + " ".join(f"{s[:12]:>12s}" for s in STAGES) + "   (ns per stage)")
    for w in range(nw):
        d = np.diff(st[w, :10]) / (cyc / wall)
        print(f"   {w:4d} {(st[w, 0] - t0) / (cyc / wall):6.0f} " + " ".join(f"{v:12.0f}" for v in d))
