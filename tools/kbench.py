#!/usr/bin/env python3
"""Kernel-level A/B of the filter_scan_kernel tiling probes (option "filter_variant"), one process,
interleaved rounds, kernel-exact durations from dispatch-attached HIP events (moihgp_profile_*).
The probes exist only in a tuning build:  make -C multioutputihgp_amd/csrc TUNING=1  and  MOIHGP_LIB=multioutputihgp_amd/lib/libmoihgp_tuning.so.
usage: python tools/kbench.py [--dtype f32|f64] [--L 4096] [--T 10000] [--variants 0,1,2,...] [--rounds 5] [--nan 0.0]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from bench import synth_params, synth_stream, SEED
from multioutputihgp_amd.streams import LatentBank

ap = argparse.ArgumentParser()
ap.add_argument("--dtype", default="f32"); ap.add_argument("--L", type=int, default=4096); ap.add_argument("--T", type=int, default=10000)
ap.add_argument("--variants", default="0,1,2,3,4,5"); ap.add_argument("--rounds", type=int, default=5); ap.add_argument("--per", type=int, default=10)
ap.add_argument("--nan", type=float, default=0.0); ap.add_argument("--mode", default="fn", help="f=yhat, n=nll")
ap.add_argument("--cold", action="store_true", help="evict L2 / Infinity Cache before every launch (1 GiB write)")
ap.add_argument("--tiled", action="store_true", help="segment-major streams (moihgp_filter_stream_tiled) next to the series-major ones: variant -1")
ap.add_argument("--rotate", type=int, default=0, help="rotate every launch over N distinct (input, output) stream pairs (bench.py's cold leg uses >= 768 MiB of them)")
a = ap.parse_args()
dtype = torch.float32 if a.dtype == "f32" else torch.float64
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
prm = synth_params(a.L, 0, np.random.default_rng(SEED))
bank = LatentBank(0.1, prm, kernel="Matern52ss")
Ty = synth_stream(a.L, 0, a.T, dtype, dev, SEED + 1)
if a.nan > 0:
    Ty[torch.rand(Ty.shape, device=dev) < a.nan] = float("nan")
yhat = torch.empty_like(Ty); nll = torch.empty((a.L,), dtype=torch.float64, device=dev)
x = torch.zeros((a.L, 3), dtype=dtype, device=dev)
variants = [int(v) for v in a.variants.split(",")]
evict = torch.zeros(256 * 1024 * 1024, dtype=torch.float32, device=dev) if a.cold else None
rot = [(Ty, yhat)] + [(Ty.clone(), torch.empty_like(yhat)) for _ in range(max(0, a.rotate - 1))]
if a.tiled:
    from multioutputihgp_amd.streams import tile_stream, untile_stream
    rot_t = [(tile_stream(t, a.T), torch.empty_like(tile_stream(t, a.T))) for t, _ in rot]
    variants = variants + [-1] + [-v for v in variants if 20 <= v < 30]     # -1: the tiled sweep as shipped; -2x: tiled with ring probe 2x
launch_no = 0
es = 4 if dtype == torch.float32 else 8
nbytes = (("f" in a.mode) + 1) * es * a.L * a.T
times = {v: [] for v in variants}; ref = None
bank.profile_enable(a.per)
for rnd in range(a.rounds):
    for v in variants:
        bank.set_option("filter_variant", v if v >= 0 else (0 if v == -1 else -v))
        for _ in range(a.per):
            if a.cold:
                evict.add_(1.0)
            x.zero_()
            if v < 0:
                ty_k, yh_k = rot_t[launch_no % len(rot_t)]; launch_no += 1
                bank.filter_tiled(ty_k, a.T, x=x, yhat=yh_k, nll=nll, want_yhat="f" in a.mode, want_nll="n" in a.mode)
                continue
            ty_k, yh_k = rot[launch_no % len(rot)]; launch_no += 1
            bank.filter(ty_k, T=a.T, x=x, yhat=yh_k, nll=nll, want_yhat="f" in a.mode, want_nll="n" in a.mode)
        times[v] += bank.profile_read()
        torch.cuda.synchronize()
        if "n" in a.mode:
            tot = nll.sum().item()
            if ref is None: ref = tot
            assert v == 9 or abs(tot - ref) <= 1e-5 * abs(ref), (v, tot, ref)
print(f"dtype={a.dtype} L={a.L} T={a.T} mode={a.mode} nan={a.nan} cold={a.cold} rotate={a.rotate} bytes/launch={nbytes/1e6:.1f} MB", flush=True)
for v in variants:
    t = np.array(times[v]) * 1e3
    print(f"variant {v}: min {t.min():7.1f} us  median {np.median(t):7.1f} us  -> {nbytes/np.median(t)/1e6:6.2f} TB/s ({nbytes/np.median(t)/1e6/8*100:4.1f}% of 8 TB/s)")
