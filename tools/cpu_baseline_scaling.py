#!/usr/bin/env python3
"""The fair CPU baseline (oracle orc_filter_stream_fast, -O3 -march=native) against the number of OpenMP threads on this host: C3 shape, fp32.
usage: python tools/cpu_baseline_scaling.py [threads ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import cref
from bench import synth_params, SEED
L, T = 4096, 10000
prm = synth_params(L, 0, np.random.default_rng(SEED), "Matern52ss")
igps = cref.ihgp_array("Matern52", 0.1, prm)
rng = np.random.default_rng(1)
Ty = (np.sin(0.05 * np.arange(T)[None, :] * (1 + np.arange(L)[:, None] % 7)) + 0.1 * rng.standard_normal((L, T))).astype(np.float32)
out = np.zeros_like(Ty)
for nt in ([int(a) for a in sys.argv[1:]] or [1, 2, 4, 8, 16, 32]):
    best = 1e9
    for _ in range(20):
        t0 = time.perf_counter(); cref.filter_stream_fast(igps, Ty, nthreads=nt, native=True, yhat_out=out); best = min(best, time.perf_counter() - t0)
    print(f"{nt:3d} threads: {L * T / best / 1e9:7.3f} Gsteps/s   {best / (L * T) * nt * 1e9:6.2f} ns per step and thread", flush=True)
