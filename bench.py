#!/usr/bin/env python3
"""bench.py -- Kalman steps/s of the MOIHGP hot path on MI355X (BASELINE.json metric).

A "step" of this bench = ONE pass of the per-latent infinite-horizon Kalman recursion + NLL over one
batch: L latents x T ticks of a pre-projected, HBM-resident, series-major stream (filtered means written,
per-latent NLL accumulated) -- moihgp_filter_stream() of include/moihgp.h.  One "Kalman step" (the metric's
unit) = one latent x one tick.  value = N * L * T / seconds_per_pass, whole job.

Default workload (N=1): BASELINE.json's target configuration "M=4096 outputs, T=10000, Matern-5/2, fp32,
1xMI355X" (configs[2] without its L-BFGS outer loop, which stays on the host).  N>1: every rank owns 4096
latents of a 4096*N-output model (weak scaling, configs[3] at N=8); the only collective is the RCCL
all-reduce of the scalar NLL.  Other configs: --config c1 | c2 | c3f64 | c2d6 | c5 | c4 | c3learn (configs[2] with its outer loop: one
objective evaluation of the online learner, MOIHGP::update + the windowed NLL/gradient sweep, at M = L = 4096).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config c3|c1|c2|c3f64|c2d6|c3d6|c3d6f64|c5|c4|c3learn|c3grad|c5grad] [--no-cpu]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

`python bench.py --gpus N` with N > 1 and no launcher around it (WORLD_SIZE unset) starts the N ranks itself: the parent, which makes
no GPU call, runs `python -m torch.distributed.run --nproc-per-node N bench.py ...` as a CHILD process (never exec in place), relays
rank 0's JSON line and exits with the children's return code.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist

# `value` / `ms_per_step` come from a timed region WITHOUT instrumentation: a HIP event pair keeps its dispatch from overlapping the
# tail of the launch before it (3-6 us per pass: +4 % at C3, +17 % at C2).  The kernel duration the roofline needs is read in a SECOND
# loop of KERNEL_PASSES passes behind the timed region, every launch bracketed by an event pair attached to the dispatch itself
# (a bracketed launch that follows un-bracketed ones reads 1-2 us long, so all of them are bracketed there).
PROFILE_STRIDE = 1
KERNEL_PASSES = 50

CONFIGS = {
    # name: (L per GPU, T, dtype, kernel, description)
    "c3": (4096, 10000, torch.float32, "Matern52ss", "C3-filter: M=L=4096/GPU, T=10000, Matern-5/2 (matern52ss.h, d=3), fp32, filter+NLL"),
    "c3f64": (4096, 10000, torch.float64, "Matern52ss", "C3 shape in fp64: M=L=4096/GPU, T=10000, Matern-5/2 (d=3), fp64, filter+NLL"),
    "c2": (256, 10000, torch.float64, "Matern52ss", "C2: M=L=256, T=10000, Matern-5/2 (d=3), fp64, fixed hyper-parameters, filter+NLL"),
    # stacked state (sum of J Matern-5/2 components): BASELINE.json's d=6 / d=12 shapes; not models of the reference
    "c2d6": (256, 10000, torch.float64, "Matern52x2", "C2 as BASELINE.json words it: M=L=256, T=10000, 2 stacked Matern-5/2 (d=6), fp64, filter+NLL"),
    # the north-star's "d~6 Matern-5/2" reading of the target shape (M = 4096, T = 10^4), in the target's fp32 and in the reference's fp64
    "c3d6": (4096, 10000, torch.float32, "Matern52x2", "C3 shape at d=6: M=L=4096, T=10000, 2 stacked Matern-5/2 (d=6), fp32, filter+NLL"),
    "c3d6f64": (4096, 10000, torch.float64, "Matern52x2", "C3 shape at d=6 in fp64: M=L=4096, T=10000, 2 stacked Matern-5/2 (d=6), fp64, filter+NLL"),
    "c5": (4096, 10000, torch.float64, "Matern52x4", "C5: M=L=4096, T=10000, 4 stacked Matern-5/2 (d=12), fp64, filter+NLL (VALU-bound)"),
    # one GPU's shard of BASELINE.json configs[3] (32768 latents over 8 GPUs, T = 1e5): the stream is swept in 1e4-tick slabs that
    # carry the state, the way a stream that does not fit would be fed
    "c4": (4096, 100000, torch.float32, "Matern52ss", "C4 shard: 4096 latents/GPU (32768 over 8), T=100000 in 10 slabs of 10000 ticks, Matern-5/2 (d=3), fp32, filter+NLL"),
}
SLAB = {"c4": 10000}           # ticks per launch; configs not listed are swept in one launch
ORACLE_KERNEL = {"Matern52ss": "Matern52", "Matern32": "Matern32"}      # product kernel name -> oracle kernel name
HBM_PEAK_GBPS = 8000.0        # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
HBM_COPY_GBPS = 6290.0
SEED = 20260101


def synth_params(L, lo, rng, kernel="Matern52ss"):
    # mag~U(0.5,2), l~U(0.5,2), noise~U(0.05,0.2); drawn for the GLOBAL latent index range
    J = int(kernel[-1]) if "x" in kernel else 1
    cols = []
    for _ in range(J):
        cols += [rng.uniform(0.5, 2, L), rng.uniform(0.5, 2, L)]
    return np.column_stack(cols + [rng.uniform(0.05, 0.2, L)])


def filter_kernel_name(bank, L, T, dtype):
    """Name of the kernel the library's own selection takes for this sweep (csrc/recursion_x.hip launch_xd, csrc/capi.cpp): a label for the
    JSON line, checked against the rocprofv3 summaries under profiles/."""
    few = L <= 256 and 1024 < T <= 16384
    fp32 = dtype == torch.float32
    if bank.stacked:
        if few and not (fp32 and bank.d > 6):
            return "filter_x_teamc_kernel" if (fp32 or bank.d <= 8) else "filter_x_team_kernel"
        return "filter_x_kernel"
    if few and bank.d == 3:
        return "filter_x_teamc_kernel"
    return "filter_dma_kernel" if L > 512 else "filter_scan_kernel"     # (many latents: the LDS-DMA sweep; up to 512: the time split of recursion.hip)


def synth_stream(L, lo, T, dtype, device, seed):
    """Ty[l][t] = sin(0.05 t (1 + l mod 7)) + 0.1 N(0,1), generated on the device."""
    from multioutputihgp_amd.streams import alloc_stream
    g = torch.Generator(device=device); g.manual_seed(seed)
    Ty = alloc_stream(L, T, dtype, device)
    Ty.zero_()
    t = torch.arange(T, device=device, dtype=torch.float64)[None, :]
    l = (torch.arange(L, device=device) + lo)[:, None]
    clean = torch.sin(0.05 * t * (1 + l % 7).to(torch.float64))
    Ty[:, :T] = (clean + 0.1 * torch.randn((L, T), generator=g, device=device, dtype=torch.float64)).to(dtype)
    return Ty


def cpu_baseline(prm, Ty_host, T, nll_gpu_sum, yhat_gpu_sub, sub, kernel="Matern52ss"):
    """Time the CPU restatement (oracle/, kind 'port': the reference itself cannot be built here) on this
    box's host cores, on the same workload; also returns the parity figures of the metric."""
    from oracle import cref
    okern = ORACLE_KERNEL.get(kernel, kernel)
    wide = cref.is_wide(okern)
    native = True
    try:
        cref.build(native=True, wide=wide)
    except Exception:
        native = False
    Lb = cref.lib(native, wide)
    cores = usable_cpus()
    nthreads = max(1, min(cores, int(Lb.orc_max_threads())))
    L = prm.shape[0]
    igps = cref.ihgp_array(okern, 0.1, prm, native=native)
    f32 = Ty_host.dtype == np.float32
    # (ii) fair-optimised (round 4: orc_filter_stream_fast -- state dimension fixed at compile time, layout hoisted, no division or isnan
    # branch per tick, 8 / 16 latents side by side in a SIMD register, OpenMP over blocks of latents on all cores, -O3 -march=native);
    # bounded to ~10 s of CPU work.  The generic loop it replaces as the headline baseline is timed next to it.
    def best_of(fn, budget):
        reps, t_best = 0, 1e30
        t_end = time.perf_counter() + budget
        while reps < 3 or (time.perf_counter() < t_end and reps < 200):
            t0 = time.perf_counter()
            fn()
            t_best = min(t_best, time.perf_counter() - t0)
            reps += 1
        return t_best, reps
    Ty_host = np.ascontiguousarray(Ty_host)
    yh_buf = np.zeros_like(Ty_host)      # the outputs land in ONE buffer, touched before the clock starts (round 4: a fresh 164 MB array per pass cost
                                         # 15-20 ms of page faults -- five times the sweep itself on 16 threads, and all of the reported time)
    t_best, reps = best_of(lambda: cref.filter_stream_fast(igps, Ty_host, x0=None, want_yhat=True, nthreads=nthreads, native=native, yhat_out=yh_buf), 10.0)
    v_opt = L * T / t_best
    t_gen, reps_gen = best_of(lambda: cref.filter_stream(igps, Ty_host, x0=None, want_yhat=True, nthreads=nthreads, native=native, yhat_out=yh_buf), 6.0)
    del yh_buf
    ghz = cpu_ghz()
    # (i) reference-shaped: single thread, one call per (tick, latent), heap temporaries (fp64 like the reference)
    import ctypes as C
    Ls = min(L, 512)
    Ty64 = np.ascontiguousarray(Ty_host[:Ls].astype(np.float64))
    x = np.zeros((Ls, int(igps[0].d))); yh = np.zeros_like(Ty64)
    dp = C.POINTER(C.c_double)
    t0 = time.perf_counter()
    Lb.orc_filter_stream_refshaped(igps, Ls, T, Ty64.ctypes.data_as(dp), Ty64.shape[1], 0, x.ctypes.data_as(dp), yh.ctypes.data_as(dp))
    t_ref = time.perf_counter() - t0
    v_ref = Ls * T / t_ref
    # parity figures of the metric: fp64 oracle on identical inputs
    o64 = cref.filter_stream(igps, np.ascontiguousarray(Ty_host.astype(np.float64)), nthreads=nthreads, native=native)
    nll_rel = abs(nll_gpu_sum - o64["nll"]) / abs(o64["nll"])
    mean_rel = float(np.max(np.abs(yhat_gpu_sub - o64["yhat"][sub])) / np.max(np.abs(o64["yhat"][sub])))
    return dict(
        value=v_opt, unit="Kalman steps/s", cores=nthreads, kind="port",
        sample=f"full workload L={L} x T={T} ({'fp32' if f32 else 'fp64'}), best of {reps} passes, oracle/moihgp_oracle.c orc_filter_stream_fast "
               f"({'-O3 -march=native' if native else '-O2'}: d-specialised, SIMD across latents, OpenMP over blocks of latents)",
        ns_per_step_per_thread=t_best * nthreads / (L * T) * 1e9,
        cycles_per_step_per_thread=(t_best * nthreads / (L * T) * ghz * 1e9) if ghz else None, cpu_ghz_assumed=ghz,
        host=dict(hardware_threads=os.cpu_count(), usable_by_this_process=cores, threads_used=nthreads),
        generic_loop=dict(value=L * T / t_gen, cores=nthreads, ns_per_step_per_thread=t_gen * nthreads / (L * T) * 1e9,
                          sample=f"same workload, orc_filter_stream (run-time state dimension, per-tick layout / isnan branches and division): the round-3 baseline, best of {reps_gen}"),
        reference_shaped=dict(value=v_ref, cores=1, sample=f"L={Ls} x T={T} fp64, one call per (tick, latent), heap temporaries (BASELINE.md variant i)"),
    ), nll_rel, mean_rel


def usable_cpus():
    """Host threads this process may actually run on: the scheduler affinity mask, capped by the cgroup's CPU quota (a GPU box hands a
    one-GPU job a share of its 128 hardware threads; os.cpu_count() reports the machine).  An OpenMP team wider than this only
    oversubscribes the share."""
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        n = os.cpu_count() or 1
    try:
        q = open("/sys/fs/cgroup/cpu.max").read().split()
        if q[0] != "max":
            n = min(n, max(1, int(np.ceil(int(q[0]) / int(q[1])))))
    except Exception:
        try:
            quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0:
                n = min(n, max(1, int(np.ceil(quota / period))))
        except Exception:
            pass
    return n


def cpu_ghz():
    """Clock to convert the CPU baseline's seconds into cycles: the host's maximum frequency if the kernel exposes it, else the current
    one of /proc/cpuinfo (an approximation either way: turbo and SMT sharing move it)."""
    try:
        return int(open("/sys/devices/system/cpu/cpu0/cpufreq/cpuinfo_max_freq").read()) / 1e6
    except Exception:
        pass
    try:
        mhz = [float(ln.split(":")[1]) for ln in open("/proc/cpuinfo") if ln.lower().startswith("cpu mhz")]
        return max(mhz) / 1e3 if mhz else None
    except Exception:
        return None


def run_c1(args, rank, world):
    """BASELINE.json configs[0], the reference's own CPU-runnable case (example.py / example_regression): M = L = 4, T = 500,
    Matern-3/2, fp64, driven tick by tick through the reference ABI (gp32_step3) exactly as example.py:40-42 does.  It measures
    call latency, not bandwidth: one step = one ABI call for all 4 latents."""
    from multioutputihgp_amd import MOIHGP
    M = L = 4; T = 500
    rng = np.random.default_rng(SEED)
    gp = MOIHGP(0.1, M, L, kernel="Matern32")
    params = gp.params.copy()
    params[M * L + L + 1:] = synth_params(L, 0, rng).ravel()
    gp.update(params)
    Y = np.sin(0.05 * np.arange(T)[:, None] * (1 + np.arange(M)[None, :])) + 0.1 * rng.standard_normal((T, M))

    def one_pass():
        x = np.zeros((L, gp.igp_dim)); out = np.empty((T, M))
        for t in range(T):
            x, out[t] = gp.step(x, Y[t])
        return x, out

    for _ in range(max(1, min(args.warmup, 2))):
        one_pass()
    steps = max(1, min(args.steps, 20))
    t0 = time.perf_counter()
    for _ in range(steps):
        xT, Yhat = one_pass()
    elapsed = time.perf_counter() - t0
    per_call = elapsed / (steps * T)
    out = {
        "metric": "Kalman steps/sec (M outputs x T ticks) + NLL rel-err vs CPU oracle",
        "value": L * T / (elapsed / steps), "unit": "Kalman steps/s", "n_gpus": 1, "steps": steps, "warmup": args.warmup,
        "ms_per_step": elapsed / steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "C1 (example.py shape): M=L=4, T=500, Matern-3/2 (d=2), fp64, one gp32_step3 call per tick", "latents_total": L, "ticks": T,
                   "state_dim": gp.igp_dim, "layout": "host buffers through the per-tick reference ABI"},
        "roofline": {"bound": "hbm", "achieved": 2 * 8 * L / per_call / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": 2 * 8 * L / per_call / 1e9 / HBM_PEAK_GBPS,
                     "traffic": None, "kernel": "fused_step_kernel", "kernel_ms": None, "call_us": per_call * 1e6,
                     "note": "launch-latency bound by construction (64 bytes per call); see the batched entries for throughput"},
    }
    if not args.no_cpu:
        from oracle import cref
        ref = cref.GP(0.1, M, L, "Matern32")
        ref.update(gp.params)
        x = np.zeros((L, gp.igp_dim)); ref_out = np.empty((T, M))
        t0 = time.perf_counter()
        for _ in range(20):
            x = np.zeros((L, gp.igp_dim))
            for t in range(T):
                x, ref_out[t] = ref.step(x, Y[t])
        tc = (time.perf_counter() - t0) / 20
        out["cpu_baseline"] = dict(value=L * T / tc, unit="Kalman steps/s", cores=1, kind="port",
                                   sample="full workload, 20 passes, oracle/moihgp_oracle.c through ctypes, one call per tick")
        out["filtered_mean_rel_err"] = float(np.max(np.abs(Yhat - ref_out)) / np.max(np.abs(ref_out)))
        out["speedup_vs_cpu_all_cores"] = out["value"] / out["cpu_baseline"]["value"]
    print(json.dumps(out), flush=True)


def learn_row(steps, warmup, cpu=True, windows=(16, 128)):
    """BASELINE.json configs[2] as worded: "M=4096 outputs, T=10000, Matern-5/2, fp32, online-learning L-BFGS outer loop".  The
    optimiser stays on the host (north_star); what it calls per line-search point is ONE objective evaluation
    (moihgp_online.h:40-72, online_learning.py:74-98):  MOIHGP::update(params) -- polar factor of the M x L mixing + IHGP::update of
    every latent -- then the window loop of W ticks (step with sensitivities + NLL gradient per tick).  A step of this bench = one
    such evaluation at M = L = 4096 (gp52 surface of include/moihgp.h: gpXX_update + moihgp_window_eval), fp64 like the reference's
    learner arithmetic; value = Kalman steps (L x W) per second of whole evaluations, for W in {16, 128}.  Two forms are
    timed: the reference ABI's (host parameter / gradient vectors: 2 x 134 MB over PCIe per evaluation) and the device-resident one
    (moihgp_update_dev + moihgp_window_eval_dev: parameters and gradient stay in HBM, what an optimiser that keeps theta, g and its
    history on the GPU pays)."""
    from multioutputihgp_amd import MOIHGP, load_library
    M = L = 4096
    rng = np.random.default_rng(SEED)
    gp = MOIHGP(0.1, M, L, kernel="Matern52ss")
    p = gp.params.copy()
    dU = rng.standard_normal(M * L)
    p[:M * L] += 0.1 * dU / np.linalg.norm(dU)               # ||dU||_F = 0.1: the reference's L-BFGS-B max_step (moihgp_online.h:156)
    p[M * L + L + 1:] = synth_params(L, 0, rng).ravel()
    d = gp.igp_dim
    x = np.zeros((L, d)); dx = np.zeros((L, 3, d))
    steps = max(1, min(steps, 10)); warm = max(1, min(warmup, 2))
    dev = torch.device("cuda", torch.cuda.current_device())
    have_dev = hasattr(gp, "update_dev")
    if have_dev:
        p_dev = torch.from_numpy(p).to(dev); g_dev = torch.empty_like(p_dev); loss_dev = torch.zeros(1, dtype=torch.float64, device=dev)
        x_dev = torch.zeros((L, d), dtype=torch.float64, device=dev); dx_dev = torch.zeros((L, 3, d), dtype=torch.float64, device=dev)
    rows = {}
    for W in windows:
        Y = 0.5 * rng.standard_normal((W, M))
        for _ in range(warm):
            gp.update(p); gp.window_objective(Y, x, dx)
        torch.cuda.synchronize()
        t_upd = t_ev = 0.0
        for _ in range(steps):
            t0 = time.perf_counter(); gp.update(p); t1 = time.perf_counter()
            loss, grad, _, _ = gp.window_objective(Y, x, dx, set_window=False); t2 = time.perf_counter()
            t_upd += t1 - t0; t_ev += t2 - t1
        t_upd /= steps; t_ev /= steps
        rows[W] = dict(update_ms=t_upd * 1e3, window_eval_ms=t_ev * 1e3, evaluation_ms=(t_upd + t_ev) * 1e3,
                       kalman_steps_per_s=L * W / (t_upd + t_ev), loss=float(loss))
        if have_dev:
            for _ in range(warm):
                gp.update_dev(p_dev); gp.window_objective_dev(x_dev, dx_dev, loss_dev, g_dev)
            torch.cuda.synchronize()
            t_upd = t_ev = 0.0
            for _ in range(steps):
                t0 = time.perf_counter(); gp.update_dev(p_dev); torch.cuda.synchronize(); t1 = time.perf_counter()
                gp.window_objective_dev(x_dev, dx_dev, loss_dev, g_dev); torch.cuda.synchronize(); t2 = time.perf_counter()
                t_upd += t1 - t0; t_ev += t2 - t1
            t_upd /= steps; t_ev /= steps
            rows[W]["device_resident"] = dict(update_ms=t_upd * 1e3, window_eval_ms=t_ev * 1e3, evaluation_ms=(t_upd + t_ev) * 1e3,
                                              kalman_steps_per_s=L * W / (t_upd + t_ev), loss=float(loss_dev.item()),
                                              grad_max_abs_diff_vs_host_path=float((g_dev.cpu().numpy() - grad).__abs__().max()))
    W = windows[-1]
    # roofline of the dominant part: the fp64 MFMA GEMMs of the polar factor (Newton-Schulz: per step one symmetric Gram 2 M L^2 / 2
    # upper tiles + one product 2 M L^2) and of the window (projection, U U^T y, U-gradient: 3 x 2 W M L)
    ns_steps = max(1, int(load_library().moihgp_polar_iterations(gp.handle)))   # what the last update() actually took
    flops = ns_steps * (1.0 * M * L * L + 2.0 * M * L * L) + 3 * 2.0 * W * M * L
    best = rows[W]["device_resident"] if have_dev else rows[W]
    t = best["evaluation_ms"] * 1e-3
    out = {
        "metric": "Kalman steps/sec (M outputs x T ticks) + NLL rel-err vs CPU oracle",
        "value": best["kalman_steps_per_s"], "unit": "Kalman steps/s", "n_gpus": 1, "steps": steps, "warmup": warm,
        "ms_per_step": best["evaluation_ms"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"C3-learn: one objective evaluation of the online learner at M=L=4096, Matern-5/2 (d=3), window W={W}: "
                               "gp52_update (device polar factor + IHGP::update x 4096) + moihgp_window_eval (projection, sensitivity sweep, NLL gradient)"
                               + ("; parameters and gradient device-resident (moihgp_update_dev / moihgp_window_eval_dev)" if have_dev else ""),
                   "latents_total": L, "outputs": M, "window": W, "state_dim": d, "gradient_entries": int(gp.num_param)},
        "roofline": {"bound": "mfma", "achieved": flops / t / 1e12, "peak": 78.6, "unit": "TFLOP/s", "frac": flops / t / 1e12 / 78.6, "traffic": None,
                     "kernel": "gemm_mfma (polar factor Newton-Schulz + window GEMMs)", "kernel_ms": None,
                     "update_ms": best["update_ms"], "window_eval_ms": best["window_eval_ms"],
                     "host_path_evaluation_ms": rows[W]["evaluation_ms"],
                     "note": f"flops = {ns_steps} Newton-Schulz steps x (M L^2 Gram upper tiles + 2 M L^2 product) + 3 window GEMMs of 2 W M L, over the WALL time of a whole "
                             "evaluation" + (" (device-resident form; host_path_evaluation_ms = the reference ABI's form, 2 x 134 MB over PCIe included)" if have_dev
                                             else " (host staging of the 134 MB parameter / gradient vectors over PCIe included)") + "; fp64 MFMA dense peak"},
        "windows": {str(k): v for k, v in rows.items()},
    }
    if cpu:
        # the oracle's own loop on a stated subsample: the reference's form is O(M^3 L^2) per tick (literal U-gradient) and its polar
        # factor an SVD; the oracle runs the closed-form U-gradient and a one-sided Jacobi SVD on 1 core
        from oracle import cref
        Ms = Ls = 256; Ws = 16
        ref = cref.GP(0.1, Ms, Ls, "Matern52"); ref.set_literal_ugrad(0)
        ps = np.concatenate([(np.eye(Ms, Ls) + 0.01 * rng.standard_normal((Ms, Ls))).ravel(), np.ones(Ls), [0.01], synth_params(Ls, 0, rng).ravel()])
        Ys = 0.5 * rng.standard_normal((Ws, Ms))
        t0 = time.perf_counter()
        ref.update(ps)
        xs, dxs = np.zeros((Ls, 3)), np.zeros((Ls, 3, 3))
        for t_ in range(Ws):
            ref.negLogLikelihood(xs, Ys[t_], dxs)
            xs, _, dxs = ref.step(xs, Ys[t_], dxs)
        tc = time.perf_counter() - t0
        out["cpu_baseline"] = dict(value=Ls * Ws / tc, unit="Kalman steps/s", cores=1, kind="port",
                                   sample=f"one evaluation at M=L={Ms}, W={Ws} (update + {Ws} x (negLogLikelihood with gradient + step)), oracle/moihgp_oracle.c through ctypes, "
                                          "closed-form U-gradient; the GPU line is M=L=4096")
        out["speedup_vs_cpu_all_cores"] = None
    del gp
    return out


def loop_row(ticks=3, W=16, threading=True):
    """BASELINE.json configs[2] as a LOOP: ticks of the online learner at M = L = 4096 (moihgp_online.h:173-187: filter the new observation, then
    re-fit on the window, <= 5 L-BFGS iterations of <= 20 line-search points each), with theta, its gradient and the optimiser's correction
    pairs on the device: the Eigen-free C++ learner of include/moihgp_cxx/lbfgsb_dev.hpp (tools/cxx/learner_bench.cpp, built here with g++
    against libmoihgp.so).  `threading` = the reference's constructor flag: with it off (the reference's default) the value the objective
    returns lacks the per-latent losses its gradient belongs to (moihgp.h:590 vs :597-607) and the line search exhausts its 20 points."""
    import subprocess
    exe = os.path.join(ROOT, "build", "learner_bench")
    libdir = os.path.join(ROOT, "multioutputihgp_amd", "lib")
    os.makedirs(os.path.dirname(exe), exist_ok=True)
    subprocess.run(["g++", "-std=c++14", "-O2", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tools", "cxx", "learner_bench.cpp"), "-o", exe,
                    "-L", libdir, "-lmoihgp", f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib"], check=True)
    M = L = 4096
    r = subprocess.run([exe, str(M), str(L), str(W), str(ticks), "1", "1" if threading else "0"], capture_output=True, text=True, timeout=900)
    if r.returncode != 0:
        raise RuntimeError(f"learner_bench failed ({r.returncode}): {r.stderr[-500:]}")
    d = json.loads(r.stdout.strip().splitlines()[-1])
    sec = d["seconds_per_tick"]
    ev = d["objective_evaluations_per_tick"]
    return {
        "metric": "Kalman steps/sec (M outputs x T ticks) + NLL rel-err vs CPU oracle",
        "value": L * W * ev / sec, "unit": "Kalman steps/s", "n_gpus": 1, "steps": ticks, "warmup": 1, "ms_per_step": sec * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"C3-loop: ticks of the online learner at M=L=4096, Matern-5/2 (d=3), window W={W}, threading={'on' if threading else 'off'}: step + "
                               "L-BFGS re-fit (<= 5 iterations x <= 20 line-search points), parameters / gradient / correction pairs device-resident",
                   "latents_total": L, "outputs": M, "window": W, "gradient_entries": d["num_param"]},
        "roofline": {"bound": "mfma", "achieved": None, "peak": 78.6, "unit": "TFLOP/s", "frac": None, "traffic": None,
                     "kernel": "gemm_mfma (polar factor) + window pipeline + vecops", "kernel_ms": None,
                     "note": "wall time of whole learner ticks; see c3learn for the roofline of one objective evaluation"},
        "learner": d, "ms_per_objective_evaluation_incl_optimiser": sec * 1e3 / max(ev, 1e-9),
    }


def grad_row(config, steps, warmup, cpu=True):
    """Mode G of SURVEY 8(d) over whole streams: the sensitivity / gradient sweep (ihgp.h:37-57 + :212-222 per tick: step with
    sensitivities, NLL and its gradient w.r.t. the latent's hyper-parameters), resident streams, one GPU.  c3grad: C3's shape
    (4096 x 10^4, Matern-5/2, d = 3, P = 3, fp32); c5grad: C5's (d = 12 stacked, P = 9, fp64).  A step = one sweep."""
    from multioutputihgp_amd.streams import LatentBank
    L, T, dtype, kernel, _ = CONFIGS["c3" if config == "c3grad" else "c5"]
    dev = torch.device("cuda", torch.cuda.current_device())
    prm = synth_params(L, 0, np.random.default_rng(SEED), kernel)
    bank = LatentBank(0.1, prm, kernel=kernel)
    Ty = synth_stream(L, 0, T, dtype, dev, SEED + 1)
    d, P = bank.d, bank.P
    for _ in range(max(1, warmup)):
        r = bank.grad(Ty, T=T, want_yhat=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        r = bank.grad(Ty, T=T, want_yhat=False)
    torch.cuda.synchronize()
    sec = (time.perf_counter() - t0) / steps
    steps_per_s = L * T / sec
    peak = 157.3 if dtype == torch.float32 else 78.6
    ref_flops = 2 * d * d + 4 * d + 5 + P * (4 * d * d + 6 * d + 8)              # SURVEY 8(d), mode G, the reference's dense form
    if bank.stacked:
        # what grad_scan_x.hip's innovation form needs at least (A block diagonal, dA_p one block or none): x' = A x + K v and
        # v = y - HA x: d DB + 2 d;  per parameter dz' = A dz + dK v + K dv, dv = -HA dz: d DB + 3 d, + DB^2 + DB for a lengthscale
        J = (P - 1) // 2; DB = d // J
        min_fma = d * DB + 2 * d + P * (d * DB + 3 * d) + J * (DB * DB + DB)
        flops, form = 2 * min_fma, "innovation form with block-diagonal A (the multiply-adds grad_scan_x.hip cannot avoid; replays, scans and response sums on top are overhead)"
    else:
        flops, form = ref_flops, "SURVEY 8(d) mode G count of the reference's form"
    out = {
        "metric": "Kalman steps/sec (M outputs x T ticks) + NLL rel-err vs CPU oracle",
        "value": steps_per_s, "unit": "Kalman steps/s", "n_gpus": 1, "steps": steps, "warmup": warmup, "ms_per_step": sec * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32" if dtype == torch.float32 else "f64", "data": "synthetic",
        "config": {"workload": f"{config}: gradient sweep (step with sensitivities + NLL + dNLL/dtheta per tick) over L={L} latents x T={T} ticks, {kernel}, "
                               f"d={d}, P={P}; streams resident in HBM", "latents_total": L, "ticks": T, "state_dim": d, "hyper_parameters_per_latent": P},
        "roofline": {"bound": "valu", "achieved": steps_per_s * flops / 1e12, "peak": peak, "unit": "TFLOP/s", "frac": steps_per_s * flops / 1e12 / peak,
                     "traffic": None, "kernel": "grad_scan_x_kernel" if bank.stacked else "grad_scan_kernel", "kernel_ms": sec * 1e3,
                     "kernel_ms_from": f"wall clock over {steps} sweeps (one launch of the sweep kernel + the small continuation kernel each)",
                     "flops_per_step": flops, "flop_count": form, "reference_form_flops_per_step": ref_flops,
                     "reference_form_equivalent_TFLOPs": steps_per_s * ref_flops / 1e12,
                     "note": "vector-ALU bound (no MFMA: per-latent matrices, d <= 12); HBM side: " + f"{steps_per_s * (4 if dtype == torch.float32 else 8) / 1e9:.0f} GB/s of 8000"},
        "nll_total": float(r["nll"].sum()),
    }
    if cpu:
        from oracle import cref
        okern = ORACLE_KERNEL.get(kernel, kernel)
        wide = cref.is_wide(okern)
        cref.build(native=True, wide=wide)
        Lb = cref.lib(True, wide)
        nthreads = max(1, min(usable_cpus(), int(Lb.orc_max_threads())))
        Ls = min(L, 256 if bank.stacked else 1024)
        igps = cref.ihgp_array(okern, 0.1, prm[:Ls], native=True)
        Th = np.ascontiguousarray(Ty[:Ls, :T].double().cpu().numpy())
        t0 = time.perf_counter()
        o = cref.grad_stream(igps, Th, want_yhat=False, nthreads=nthreads)
        tc = time.perf_counter() - t0
        out["cpu_baseline"] = dict(value=Ls * T / tc, unit="Kalman steps/s", cores=nthreads, kind="port",
                                   sample=f"first {Ls} latents x T={T}, fp64, oracle/moihgp_oracle.c orc_grad_stream (-O3 -march=native, OpenMP over latents), one pass")
        g = r["grad"][:Ls].cpu().numpy(); n = r["nll"][:Ls].cpu().numpy()
        out["nll_rel_err"] = float(np.abs(n - o["nll_per_latent"]).max() / np.abs(o["nll_per_latent"]).max())
        out["grad_rel_err"] = float(np.abs(g - o["grad"]).max() / np.abs(o["grad"]).max())
        out["speedup_vs_cpu_all_cores"] = steps_per_s / out["cpu_baseline"]["value"]
    del bank, Ty
    return out


def valu_side(d, dtype, steps_per_s):
    """Vector-ALU side of the roofline: SURVEY 8(d)'s flop count for the filter (2 d^2 + 2 d per Kalman step) over the
    dense vector peak of the dtype (MI355X_MICROARCH.md: 157.3 TFLOP/s fp32, 78.6 fp64)."""
    flops = 2 * d * d + 2 * d
    peak = 157.3 if dtype == torch.float32 else 78.6
    tf = steps_per_s * flops / 1e12
    return {"flops_per_step": flops, "achieved": tf, "peak": peak, "unit": "TFLOP/s", "frac": tf / peak}


def cold_leg(bank, Ty, yhat, x, x_zero, nll, T, alg_bytes, es, tiled=False):
    """Cold-stream figure of a filter configuration: the timed passes sweep ONE resident stream, so between passes part of its input is
    served by the 256 MiB Infinity Cache (and FETCH_SIZE counts those hits): roofline.frac is cache-assisted whenever the input fits.
    Here the same launch rotates over enough distinct (input, output) pairs that nothing it reads can still be on chip."""
    pair_bytes = 2 * es * Ty.numel()
    # enough pairs that the INPUT streams alone are 3 x the 256 MiB Infinity Cache (the outputs leave through streaming stores and need not
    # occupy it; with the pairs' total as the yardstick -- 3 pairs at C3 -- part of the input was still served on chip: 66 us against 70 with 5)
    in_bytes = es * Ty.numel()
    nrot = max(2, min(10, int((3 * 256 * 2 ** 20 + in_bytes - 1) // in_bytes)))

    def sweep(a, b):
        if tiled:
            bank.filter_tiled(a, T, x=x, x_start=x_zero, yhat=b, nll=nll)
        else:
            bank.filter(a, T=T, x=x, x_start=x_zero, yhat=b, nll=nll)
    try:
        rot = [(Ty, yhat)] + [(Ty.clone(), torch.empty_like(yhat)) for _ in range(nrot - 1)]
        for k in range(nrot):
            sweep(*rot[k])
        bank.profile_enable(3 * nrot)
        for k in range(3 * nrot):
            sweep(*rot[k % nrot])
        cold_ms = float(np.mean(bank.profile_read()))
        del rot
        return {"frac_cold": alg_bytes / (cold_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, "achieved_cold": alg_bytes / (cold_ms * 1e-3) / 1e9,
                "kernel_ms_cold": cold_ms,
                "cold_note": f"same launch rotating over {nrot} distinct stream pairs (inputs {nrot * in_bytes / 2 ** 20:.0f} MiB"
                             + (" >= 3 x the 256 MiB Infinity Cache)" if nrot * in_bytes >= 3 * 256 * 2 ** 20 else ", capped at 10 pairs)")}
    except torch.cuda.OutOfMemoryError:
        return {"frac_cold": None}


FILTER_ONLY = {"c2", "c2d6"}      # BASELINE.json words configs[1] "fixed hyperparams (filter only)": its pass is the sweep; the others model objective passes


def filter_row(name, device, passes=20, warm=150):
    """One of the other single-GPU filter configurations of BASELINE.json, measured the way the headline is: kernel-exact HIP event
    pairs on every launch, the wall clock over the same passes, and the cold-stream leg.  A pass of the headline -- and of the rows that
    model an optimiser's objective passes -- is the sweep plus the one-workgroup kernel that totals the per-latent NLLs (the scalar the
    all-reduce carries); the "filter only" configuration of BASELINE.json (c2, and its d = 6 reading c2d6) is the sweep alone.  Both forms
    are timed for every row: the sweep kernel reads ~2 us longer when the tiny total kernel runs between two sweeps (rocprofv3's kernel
    trace shows the same: 255 of 256 CUs idle for 3-4 us, and the next dispatch ramps up again)."""
    from multioutputihgp_amd.streams import LatentBank
    L2, T2, dt2, k2, desc2 = CONFIGS[name]
    b2 = LatentBank(0.1, synth_params(L2, 0, np.random.default_rng(SEED), k2), kernel=k2)
    Ty2 = synth_stream(L2, 0, T2, dt2, device, SEED + 1)
    yh2 = torch.empty_like(Ty2); n2 = torch.empty((L2,), dtype=torch.float64, device=device)
    x2 = torch.zeros((L2, b2.d), dtype=dt2, device=device)
    x2z = torch.zeros_like(x2)
    tot2 = torch.zeros((1,), dtype=torch.float64, device=device)
    es2 = 4 if dt2 == torch.float32 else 8
    alg = 2 * es2 * L2 * T2
    # warm-up: the fp64 stacked kernels start slow and settle over their first ~100 launches (tools/micro/launch_dist.py: d = 6 fp64 151 us at
    # launch 1, 186 us around launch 20, 145 us from launch ~120 on -- the device's power management, not the kernel: the stream and the
    # code are the same); the rows below are steady-state figures, like a learner's repeated objective evaluations
    for _ in range(warm):
        b2.filter(Ty2, T=T2, x=x2, x_start=x2z, yhat=yh2, nll=n2, nll_total=tot2)

    def timed(with_total, bracket):
        """wall time per pass without instrumentation (bracket = False), or the mean kernel duration with an event pair on every launch"""
        b2.profile_enable(passes if bracket else 0)
        torch.cuda.synchronize()
        tw0 = time.perf_counter()
        for _ in range(passes):
            b2.filter(Ty2, T=T2, x=x2, x_start=x2z, yhat=yh2, nll=n2, nll_total=tot2 if with_total else None)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - tw0) / passes
        return float(np.mean(b2.profile_read())) if bracket else wall

    sweep_only = name in FILTER_ONLY
    timed(sweep_only, True)                                      # (the other form first, so that the reported one is measured last, warm)
    ms_o = timed(sweep_only, True)                               # other form: with the total iff the row's own pass is the sweep alone
    timed(not sweep_only, True)
    ms2 = timed(not sweep_only, True)
    wall2 = timed(not sweep_only, False)                         # the row's `value`: no event pairs
    cold = cold_leg(b2, Ty2, yh2, x2, x2z, n2, T2, alg, es2)
    b2.profile_enable(0)
    ms_head = cold["kernel_ms_cold"] if cold.get("frac_cold") is not None else ms2
    row = {"workload": desc2, "state_dim": b2.d, "dtype": "f32" if dt2 == torch.float32 else "f64", "warmup": warm, "steps": passes,
           "pass": "sweep only (BASELINE.json: filter only)" if sweep_only else "sweep + the pass's NLL total (one-workgroup kernel behind it), as the headline",
           "ms_per_step": wall2 * 1e3, "value": L2 * T2 / wall2,
           "kernel": filter_kernel_name(b2, L2, T2, dt2), "kernel_ms": ms_head,
           "steps_per_s_kernel_only": L2 * T2 / (ms_head * 1e-3), "bound": "hbm",
           "achieved_GBps": alg / (ms_head * 1e-3) / 1e9, "frac": alg / (ms_head * 1e-3) / 1e9 / HBM_PEAK_GBPS,
           "frac_is": "cold stream (rotating pairs)" if cold.get("frac_cold") is not None else "resident stream",
           "kernel_ms_resident": ms2, "frac_resident": alg / (ms2 * 1e-3) / 1e9 / HBM_PEAK_GBPS,
           ("kernel_ms_with_total_between_sweeps" if sweep_only else "kernel_ms_sweeps_back_to_back"): ms_o,
           ("frac_with_total_between_sweeps" if sweep_only else "frac_sweeps_back_to_back"): alg / (ms_o * 1e-3) / 1e9 / HBM_PEAK_GBPS,
           "vector_alu": valu_side(b2.d, dt2, L2 * T2 / (ms2 * 1e-3))}
    if row["vector_alu"]["frac"] > row["frac_resident"]:
        row["bound"] = "valu"         # the d = 12 fp64 shape sits on the vector-ALU wall: `frac` stays the HBM figure, vector_alu.frac the binding one
    row.update(cold)
    del b2, Ty2, yh2
    return row


def gaps_row(device, name="c5", frac=0.01, passes=10, warm=3):
    """A filter configuration with a fraction of its ticks missing (NaN: ihgp.h:83-87, :204-209), next to the same sweep without gaps: wall time of
    all passes per sweep (the many-latent stacked sweep takes such streams by exact imputation since round 4; VERDICT r3 item 6 asked
    <= 4 x the gap-free sweep at 1 % missing).  Median of three runs of `passes` sweeps, as tools/filternan.py."""
    from multioutputihgp_amd.streams import LatentBank
    L2, T2, dt2, k2, desc2 = CONFIGS[name]
    b2 = LatentBank(0.1, synth_params(L2, 0, np.random.default_rng(SEED), k2), kernel=k2)
    out = {"workload": f"{desc2}; {100 * frac:g} % of the ticks missing", "state_dim": b2.d, "dtype": "f32" if dt2 == torch.float32 else "f64", "missing_fraction": frac}
    for tag, dt_ in (("", dt2), ("_f32" if dt2 == torch.float64 else "_f64", torch.float32 if dt2 == torch.float64 else torch.float64)):
        res = {}
        for label, fr in (("gap_free", 0.0), ("with_gaps", frac)):
            Ty2 = synth_stream(L2, 0, T2, dt_, device, SEED + 1)
            if fr > 0:
                g = torch.Generator(device=device); g.manual_seed(SEED + 7)
                Ty2[torch.rand(Ty2.shape, generator=g, device=device) < fr] = float("nan")
            yh2 = torch.empty_like(Ty2); n2 = torch.empty((L2,), dtype=torch.float64, device=device)
            x2 = torch.zeros((L2, b2.d), dtype=dt_, device=device); x2z = torch.zeros_like(x2)
            for _ in range(warm):
                b2.filter(Ty2, T=T2, x=x2, x_start=x2z, yhat=yh2, nll=n2)
            reps = []
            for _ in range(3):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(passes):
                    b2.filter(Ty2, T=T2, x=x2, x_start=x2z, yhat=yh2, nll=n2)
                torch.cuda.synchronize(); reps.append((time.perf_counter() - t0) / passes)
            res[label] = sorted(reps)[1] * 1e3
            del Ty2, yh2
        out[f"ms_gap_free{tag}"] = res["gap_free"]; out[f"ms_with_gaps{tag}"] = res["with_gaps"]; out[f"ratio{tag}"] = res["with_gaps"] / res["gap_free"]
    out["ms_per_step"] = out["ms_with_gaps"]; out["value"] = L2 * T2 / (out["ms_with_gaps"] * 1e-3)
    out["note"] = "wall clock of every pass of the sweep (first pass, the two imputation kernels, second pass), streams resident; round 3: 12-18 x"
    del b2
    return out


def slab_row(device, passes=5, warm=2, world=1, rank=0, reduce=None):
    """BASELINE.json configs[3] as worded, this rank's part: 4096 latents x 10^5 ticks swept in 10 slabs of 10^4 ticks that carry the state,
    the per-latent NLLs summed over the slabs and (N > 1) all-reduced once per pass.  3.3 GB working set per rank: cold by construction."""
    from multioutputihgp_amd.streams import LatentBank
    L4, T4, dt4, k4, desc4 = CONFIGS["c4"]
    slab4 = SLAB["c4"]; nsl = T4 // slab4
    prm_all = synth_params(L4 * world, 0, np.random.default_rng(SEED), k4)
    b4 = LatentBank(0.1, prm_all[rank * L4:(rank + 1) * L4], kernel=k4)
    from multioutputihgp_amd.streams import tile_stream
    slabs = [tile_stream(synth_stream(L4, rank * L4, slab4, dt4, device, SEED + 100 + 17 * rank + k), slab4) for k in range(nsl)]   # segment-major slabs
    outs = [torch.empty_like(t) for t in slabs]
    n4 = torch.empty((L4,), dtype=torch.float64, device=device); acc = torch.zeros_like(n4)
    x4 = torch.zeros((L4, b4.d), dtype=dt4, device=device); xz = torch.zeros_like(x4)

    def one():
        acc.zero_()
        for k in range(nsl):
            b4.filter_tiled(slabs[k], slab4, x=x4, x_start=xz if k == 0 else None, yhat=outs[k], nll=n4)
            acc.add_(n4)
        return reduce(acc) if reduce is not None else acc.sum()

    for _ in range(warm):
        one()
    b4.profile_enable(0)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(passes):
        tot = one()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    wall = (time.perf_counter() - t0) / passes
    b4.profile_enable(passes * nsl)
    for _ in range(passes):
        one()
    torch.cuda.synchronize()
    kms = float(np.mean(b4.profile_read()))
    b4.profile_enable(0)
    alg = 2 * 4 * L4 * slab4
    row = {"workload": desc4, "dtype": "f32", "ms_per_step": wall * 1e3, "value": world * L4 * T4 / wall, "n_gpus": world,
           "kernel_ms": kms, "kernel_ms_is": "per slab launch", "bound": "hbm", "achieved_GBps": alg / (kms * 1e-3) / 1e9,
           "frac": alg / (kms * 1e-3) / 1e9 / HBM_PEAK_GBPS, "frac_is": "cold by construction (3.3 GB working set per GPU)",
           "layout": "segment-major slabs [ceil(10^4 / SEG)][L][SEG]",
           "nll_total": float(tot.item() if hasattr(tot, "item") else tot)}
    del b4, slabs, outs
    return row


def e2e_row(device, dtype, passes=5, warm=2):
    """The whole A3 pipeline on REAL (un-projected) observations at M = L = 4096, T = 10^4 (moihgp.h:181 project, the recursion, :222-225
    un-project): Y [T][M] -> Ty = S^-1/2 U^T Y^T (MFMA GEMM) -> filter sweep -> Yhat = (U S^1/2 Tyhat)^T (MFMA GEMM).  For real data the two
    GEMMs (2 M L T flop each) are the wall, priced against the MFMA peak of the dtype; the sweep against HBM."""
    from multioutputihgp_amd import MOIHGP
    from multioutputihgp_amd.streams import LatentBank, project_stream, unproject_stream
    M = L = 4096; T = 10000
    rng = np.random.default_rng(SEED + 5)
    gp = MOIHGP(0.1, M, L, kernel="Matern52ss")
    p = gp.params.copy()
    p[M * L:M * L + L] = rng.uniform(0.5, 2.0, L); p[M * L + L] = 0.04
    p[M * L + L + 1:] = synth_params(L, 0, np.random.default_rng(SEED)).ravel()
    gp.update(p)
    bank = LatentBank.from_handle(gp)
    g = torch.Generator(device=device); g.manual_seed(SEED + 6)
    Y = torch.randn((T, M), generator=g, device=device, dtype=dtype)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    acc = np.zeros(3); wall = 0.0
    for it in range(warm + passes):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ev[0].record()
        Ty = project_stream(gp, Y)
        ev[1].record()
        yl, xe, nll = bank.filter(Ty, T=T)
        ev[2].record()
        Yhat = unproject_stream(gp, yl, T)
        ev[3].record()
        torch.cuda.synchronize()
        if it >= warm:
            wall += time.perf_counter() - t0
            acc += [ev[0].elapsed_time(ev[1]), ev[1].elapsed_time(ev[2]), ev[2].elapsed_time(ev[3])]
    acc /= passes; wall /= passes
    flops = 2.0 * M * L * T
    peak = 157.3 if dtype == torch.float32 else 78.6
    es = 4 if dtype == torch.float32 else 8
    row = {"workload": f"C3-e2e: Y [T=10^4][M=4096] -> project (GEMM) -> filter + NLL (L=4096, d=3) -> unproject (GEMM), {'fp32' if es == 4 else 'fp64'}",
           "dtype": "f32" if es == 4 else "f64", "ms_per_step": wall * 1e3, "value": L * T / wall,
           "project_ms": acc[0], "filter_ms": acc[1], "unproject_ms": acc[2],
           "project_frac_of_mfma_peak": flops / (acc[0] * 1e-3) / 1e12 / peak, "unproject_frac_of_mfma_peak": flops / (acc[2] * 1e-3) / 1e12 / peak,
           "mfma_peak_TFLOPs": peak, "filter_frac_of_hbm_peak": 2 * es * L * T / (acc[1] * 1e-3) / 1e9 / HBM_PEAK_GBPS,
           "stage_ms_from": "torch events on the launch stream around each stage (the filter figure includes its NLL allocation and the total kernel)"}
    del gp, bank, Y, Ty, yl, Yhat
    return row


def csrc_digest():
    """sha256 over the kernel sources (multioutputihgp_amd/csrc/*.hip, *.h, *.cpp, Makefile; names and contents, sorted): what
    profiles/pmc_traffic.json was collected against.  (git is not available on the GPU box: the snapshot carries no .git.)"""
    import hashlib
    d = os.path.join(ROOT, "multioutputihgp_amd", "csrc")
    h = hashlib.sha256()
    for fn in sorted(os.listdir(d)):
        if fn.endswith((".hip", ".h", ".cpp")) or fn == "Makefile":
            h.update(fn.encode()); h.update(b"\0"); h.update(open(os.path.join(d, fn), "rb").read()); h.update(b"\0")
    return h.hexdigest()


def pmc_traffic(config):
    """HBM bytes per launch from the PMC counters (tools/pmc_traffic.sh: separate --pmc FETCH_SIZE / WRITE_SIZE passes with the gfx950
    correction), as committed under profiles/: a record of an earlier profiled run of this command, not a measurement of this run
    (rocprofv3 --pmc cannot run inside the timed bench).  The record names the digest of the kernel sources it was collected against;
    if the sources have changed since, `traffic` is null and the source record says why -- a stale figure is not reported."""
    pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if not os.path.exists(pmc):
        return None, None
    try:
        rec = json.load(open(pmc))
        ent = rec.get(config, {})
        src = {"file": "profiles/pmc_traffic.json", "collected_at_commit": rec.get("_commit"), "collected": rec.get("_collected"), "summary": ent.get("source")}
        now = csrc_digest()
        if rec.get("_csrc_sha256") != now:
            src["note"] = ("STALE: multioutputihgp_amd/csrc has changed since the PMC pass (sha256 of the sources then "
                           f"{str(rec.get('_csrc_sha256'))[:16]}, now {now[:16]}): traffic withheld; re-run tools/measure_all.sh")
            src["withheld_hbm_bytes_per_launch"] = ent.get("hbm_bytes_per_launch")
            return None, src
        src["note"] = "PMC pass of an earlier run of this command on the same kernel sources (sha256 matches)"
        return ent.get("hbm_bytes_per_launch"), src
    except Exception:
        return None, None


class _StubBank:
    """BENCH_REHEARSAL=stub only (tests/test_bench_launch.py): stands in for the HIP sweep so that the rank / launch / reduction logic of
    this file runs on a box without a GPU.  It computes nothing of the path; a line produced with it says so and is not a measurement."""
    d, stacked = 3, False

    def __init__(self, L):
        self.L = L

    def filter(self, Ty, T=None, x=None, x_start=None, yhat=None, nll=None, nll_total=None, **kw):
        if nll is not None:
            nll.fill_(1.0)
        if nll_total is not None:
            nll_total.fill_(float(self.L))
        return yhat, x, nll

    def profile_enable(self, n, stride=1):
        self._n = n

    def profile_read(self):
        return [1.0]


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks as CHILD processes (torch.distributed.run), relay rank 0's JSON
    line, return the children's code.  The parent has made no GPU call (importing torch does not initialise the device), and it
    does not replace itself: on this pool an exec from a process that has touched the GPU takes the machine down."""
    import socket
    import subprocess
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True)
    line = None
    for ln in proc.stdout:
        st = ln.strip()
        if st.startswith("{") and '"metric"' in st:
            line = st                                  # rank 0's result line (the only rank that prints one)
        else:
            sys.stderr.write(ln)
    rc = proc.wait()
    if line is not None:
        print(line, flush=True)
    if rc == 0 and line is None:
        sys.stderr.write("bench.py: the ranks exited cleanly but printed no result line\n")
        rc = 1
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=None, help="untimed passes before the timed ones (default: 5 for the headline c3; 150 for the other shapes, whose fp64 "
                    "stacked kernels settle over their first ~100 launches: tools/micro/launch_dist.py)")
    ap.add_argument("--config", default="c3", choices=sorted(CONFIGS) + ["c1", "c3learn", "c3loop", "c3grad", "c5grad"])
    ap.add_argument("--layout", default="auto", choices=["auto", "series", "tiled"],
                    help="stream layout of the timed sweep: series-major [L][ld] or segment-major [T/SEG][L][SEG] (include/moihgp.h moihgp_filter_stream_tiled); "
                         "auto = tiled where the sweep supports it (the reference's own models, many latents), series otherwise")
    ap.add_argument("--rotate", action="store_true",
                    help="every pass of every loop (warm-up, timed, bracketed) sweeps the NEXT of enough distinct stream pairs that nothing it reads is still "
                         "on chip: a run whose kernel trace holds cold launches only (profiles/: the rocprofv3 summary that `roofline.frac` is checked against)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline / parity leg")
    ap.add_argument("--no-cold", action="store_true", help="skip the cold-stream leg (roofline.frac_cold): profiled runs, so that a kernel trace holds the timed launches only")
    ap.add_argument("--no-others", action="store_true", help="skip the other BASELINE configurations that the default line carries in other_configs")
    ap.add_argument("--sync-allreduce", action="store_true", help="N > 1: make every pass wait for its own NLL all-reduce (no overlap with the next sweep)")
    args = ap.parse_args()
    if args.warmup is None:
        args.warmup = 5 if args.config in ("c3", "c1", "c4", "c3learn", "c3loop") else (150 if args.config in CONFIGS else 5)

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(args)             # before anything touches the GPU

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch N ranks for --gpus N (or plain `python bench.py --gpus N`, which starts them itself)")
    # Rehearsal on a 1-GPU box: BENCH_REHEARSAL=1 lets all ranks share device 0 and moves the 8-byte exchange to
    # gloo (RCCL refuses two ranks on one device); BENCH_REHEARSAL=stub additionally replaces the sweep by a stub on the CPU
    # (launch / rank / reduction logic only, for the CPU test suite).  Never used for reported numbers.
    rehearsal = os.environ.get("BENCH_REHEARSAL", "0")
    stub = rehearsal == "stub"
    rehearsal = rehearsal in ("1", "stub")
    if stub:
        device = torch.device("cpu")
    else:
        assert torch.cuda.is_available(), "bench.py needs a GPU; there is no CPU fallback"
        dev_index = 0 if rehearsal else local_rank
        torch.cuda.set_device(dev_index)
        device = torch.device("cuda", dev_index)
    # BENCH_FORCE_DIST=1: build the process group even for one rank (a 1-rank RCCL communicator: the -m gpu suite runs the N > 1 code
    # of this file -- init with device_id, the overlapped all-reduce, max-over-ranks on the device -- on the one GPU it has)
    force_dist = os.environ.get("BENCH_FORCE_DIST", "0") == "1"
    if world > 1 or force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    from multioutputihgp_amd import sharded
    from multioutputihgp_amd.sharded import (allreduce_nll, allreduce_nll_async, allreduce_total, allreduce_total_async, max_over_ranks,
                                             run_pipelined, shard_bounds)
    if force_dist:
        sharded.FORCE_COLLECTIVES = True     # lift the world == 1 short-cuts: the collectives really run

    if args.config == "c1":
        return run_c1(args, rank, world)
    if args.config == "c3learn":
        print(json.dumps(learn_row(args.steps, args.warmup, cpu=not args.no_cpu)), flush=True)
        return 0
    if args.config in ("c3grad", "c5grad"):
        print(json.dumps(grad_row(args.config, args.steps, args.warmup, cpu=not args.no_cpu)), flush=True)
        return 0
    if args.config == "c3loop":
        out = loop_row(ticks=max(1, min(args.steps, 10)), threading=True)
        out["threading_off"] = loop_row(ticks=max(1, min(args.steps, 3)), threading=False)["learner"]
        print(json.dumps(out), flush=True)
        return 0

    Lg_per, T, dtype, kernel, desc = CONFIGS[args.config]
    slab = SLAB.get(args.config, T)
    nslab = (T + slab - 1) // slab
    Lglobal = Lg_per * world
    lo, hi = shard_bounds(Lglobal, world, rank)
    L = hi - lo
    prm_all = synth_params(Lglobal, 0, np.random.default_rng(SEED), kernel)
    prm = prm_all[lo:hi]
    if stub:
        bank = _StubBank(L)
        Ty = torch.zeros((L, 16), dtype=dtype)
    else:
        from multioutputihgp_amd.streams import LatentBank
        bank = LatentBank(0.1, prm, kernel=kernel)
        Ty = synth_stream(L, lo, T, dtype, device, SEED + 1 + rank)
    yhat = torch.empty_like(Ty)
    nll = torch.empty((L,), dtype=torch.float64, device=device)
    x = torch.zeros((L, bank.d), dtype=dtype, device=device)

    nll_acc = torch.zeros((L,), dtype=torch.float64, device=device)
    if nslab > 1:
        # each slab is its own contiguous [L][slab] buffer, as slabs of a stream arrive (measured on this shape: views into one
        # [L][1e5] array, i.e. a 400 KB row stride, 4.1 TB/s; separate slabs 4.5 TB/s; one launch over all 1e5 ticks 4.9 TB/s)
        Ty_slabs = [Ty[:, k * slab:min(T, (k + 1) * slab)].contiguous() for k in range(nslab)]
        yhat_slabs = [torch.empty_like(t) for t in Ty_slabs]

    x_zero = torch.zeros_like(x)               # every pass starts from this state; it is never written
    # stream layout of the timed sweep: segment-major where the kernel takes it (the reference's own models through the one-wavefront-per-
    # latent sweep), series-major otherwise; the other layout's kernel figures ride along in roofline.other_layout (N = 1)
    can_tile = (not stub) and (not bank.stacked) and L > 512
    tiled = can_tile if args.layout == "auto" else (args.layout == "tiled")
    if tiled and not can_tile:
        raise SystemExit(f"--layout tiled: config {args.config} has no segment-major sweep (stacked model or too few latents)")
    if tiled:
        from multioutputihgp_amd.streams import tile_stream, untile_stream
        if nslab > 1:
            Tt_slabs = [tile_stream(t, t.shape[1]) for t in Ty_slabs]; yt_slabs = [torch.empty_like(t) for t in Tt_slabs]
        else:
            Tt = tile_stream(Ty, T); yt = torch.empty_like(Tt)

    tot_ring = [torch.zeros((1,), dtype=torch.float64, device=device) for _ in range(4)]   # NLL totals of the passes in flight
    # the stream pair(s) the passes sweep: one (the default: `value` is the throughput of repeated sweeps of one HBM-resident stream), or with
    # --rotate enough distinct copies that every launch of the run is a cold one
    rot_pairs = []
    if nslab == 1 and not stub:
        rot_pairs = [(Tt, yt) if tiled else (Ty, yhat)]
        if args.rotate:
            in_b = rot_pairs[0][0].numel() * rot_pairs[0][0].element_size()
            for _ in range(max(1, min(9, int((3 * 256 * 2 ** 20 + in_b - 1) // in_b) - 1))):
                rot_pairs.append((rot_pairs[0][0].clone(), torch.empty_like(rot_pairs[0][1])))
    elif nslab == 1:
        rot_pairs = [(Ty, yhat)]
    pass_no = [0]

    def one_pass(reduce=allreduce_nll):
        if nslab == 1:
            # the library queues its own one-workgroup total behind the sweep (measured against torch's .sum() on this shape:
            # 58.7 vs 59.8-61.2 us per pass); the path's only exchange is the all-reduce of that 8-byte scalar
            tot = tot_ring[pass_no[0] % len(tot_ring)]
            a_in, a_out = rot_pairs[pass_no[0] % len(rot_pairs)]
            pass_no[0] += 1
            if tiled:
                bank.filter_tiled(a_in, T, x=x, x_start=x_zero, yhat=a_out, nll=nll, nll_total=tot)
            else:
                bank.filter(a_in, T=T, x=x, x_start=x_zero, yhat=a_out, nll=nll, nll_total=tot)
            return (allreduce_total_async if reduce is allreduce_nll_async else allreduce_total)(tot)
        nll_acc.zero_()
        for k in range(nslab):                 # slabs carry the state x from one launch to the next
            if tiled:
                bank.filter_tiled(Tt_slabs[k], Ty_slabs[k].shape[1], x=x, x_start=x_zero if k == 0 else None, yhat=yt_slabs[k], nll=nll)
            else:
                bank.filter(Ty_slabs[k], T=Ty_slabs[k].shape[1], x=x, x_start=x_zero if k == 0 else None, yhat=yhat_slabs[k], nll=nll)
            nll_acc.add_(nll)
        return reduce(nll_acc)

    def sync():
        if not stub:
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        total = one_pass()
    bank.profile_enable(0)                      # the timed region carries no instrumentation
    multi = world > 1 or force_dist
    sync()
    if multi:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    if args.sync_allreduce or not multi:
        for k in range(args.steps):
            total = one_pass()
    else:
        # every pass still ends in its own all-reduce of the NLL scalar, but the exchange of pass k runs on the communicator's
        # stream while pass k+1 sweeps (two in flight at most); all of them are complete before the clock stops
        # (sharded.run_pipelined: the same loop the 8-rank gloo test drives on the CPU)
        try:
            total = run_pipelined(args.steps, lambda: one_pass(reduce=allreduce_nll_async), max_in_flight=2)[-1]
        except RuntimeError as e:                  # a communicator without async collectives: the ordered form
            sys.stderr.write(f"async all-reduce unavailable ({e}); continuing stream-ordered\n")
            for k in range(args.steps):
                total = one_pass()
    sync()
    if multi:
        dist.barrier()
    sync()
    elapsed = time.perf_counter() - t0
    elapsed = max_over_ranks(elapsed, device="cpu" if rehearsal else device)
    # kernel-exact durations, OUTSIDE the timed region: a second loop of the same passes with a HIP event pair attached to every filter
    # dispatch (hipExtLaunchKernel, on the launch stream)
    kpasses = max(1, min(KERNEL_PASSES, args.steps))
    bank.profile_enable(kpasses * nslab, stride=PROFILE_STRIDE)
    tk0 = time.perf_counter()
    for k in range(kpasses):
        total_k = one_pass(reduce=lambda t: t)
    sync()
    ms_per_step_bracketed = (time.perf_counter() - tk0) / kpasses * 1e3
    kern_samples = bank.profile_read()
    kern_ms = float(np.mean(kern_samples))                          # mean over the launches of the bracketed loop
    bank.profile_enable(0)

    # N > 1 on real GPUs: BASELINE.json configs[3] as worded (T = 10^5 in slabs, the NLL all-reduced once per pass) rides along, every rank
    # taking part, so that a scaling line also shows the collective at work on that configuration
    c4_multi = None
    if world > 1 and not rehearsal and not stub and args.config == "c3" and not args.no_others:
        try:
            c4_multi = slab_row(device, passes=3, warm=1, world=world, rank=rank, reduce=allreduce_nll)
        except Exception as e:
            c4_multi = {"error": str(e)}

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = Lglobal * T / (elapsed / args.steps)
        es = 4 if dtype == torch.float32 else 8
        alg_bytes = 2 * es * L * min(slab, T)           # per LAUNCH; SURVEY 8d mode F: read Ty + write Tyhat = 2*s B per Kalman step
        achieved = alg_bytes / (kern_ms * 1e-3) / 1e9
        traffic, traffic_source = pmc_traffic(args.config)
        cold = {}
        other_layout = None
        if nslab == 1 and not args.no_cold and not stub:
            cold = cold_leg(bank, Tt if tiled else Ty, yt if tiled else yhat, x, x_zero, nll, T, alg_bytes, es, tiled=tiled)
            if can_tile and world == 1:
                # the same sweep in the OTHER layout: kernel durations only (resident: bracketed launches with the total between them, as the
                # headline's second loop; cold: rotating pairs)
                try:
                    if tiled:
                        o_in, o_out = Ty, yhat
                    else:
                        from multioutputihgp_amd.streams import tile_stream
                        o_in = tile_stream(Ty, T); o_out = torch.empty_like(o_in)
                    bank.profile_enable(20)
                    for _ in range(20):
                        if tiled:
                            bank.filter(o_in, T=T, x=x, x_start=x_zero, yhat=o_out, nll=nll, nll_total=tot_ring[0])
                        else:
                            bank.filter_tiled(o_in, T, x=x, x_start=x_zero, yhat=o_out, nll=nll, nll_total=tot_ring[0])
                    o_ms = float(np.mean(bank.profile_read()))
                    o_cold = cold_leg(bank, o_in, o_out, x, x_zero, nll, T, alg_bytes, es, tiled=not tiled)
                    bank.profile_enable(0)
                    other_layout = {"layout": "series-major [L][ld]" if tiled else "segment-major [T/SEG][L][SEG]",
                                    "kernel_ms_resident": o_ms, "frac_resident": alg_bytes / (o_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                                    "kernel_ms_cold": o_cold.get("kernel_ms_cold"), "frac_cold": o_cold.get("frac_cold")}
                except Exception as e:
                    other_layout = {"error": str(e)}
        # roofline.frac is the HBM-honest figure.  One resident stream pair swept again and again is partly served by the 256 MiB Infinity
        # Cache (C3's input is 164 MB), so `frac` / `achieved` / `kernel_ms` are those of the COLD leg -- the same launch rotating over
        # enough distinct stream pairs that nothing it reads is still on chip -- and the resident figures are reported next to them.  A
        # slab sweep (c4) is cold by construction (3.3 GB working set); with --no-cold the resident figure stands in and says so.
        if cold.get("frac_cold") is not None:
            head_ms, head_kind = cold["kernel_ms_cold"], "cold: " + cold["cold_note"]
        elif nslab > 1:
            head_ms, head_kind = kern_ms, f"cold by construction: {nslab} slabs of one {2 * es * L * T / 2 ** 30:.1f} GiB working set"
        elif args.rotate and len(rot_pairs) > 1:
            head_ms, head_kind = kern_ms, f"cold: --rotate, every pass of the run sweeps the next of {len(rot_pairs)} distinct stream pairs"
        else:
            head_ms, head_kind = kern_ms, "RESIDENT stream (cold leg skipped): cache-assisted when the input fits the 256 MiB Infinity Cache"
        head_achieved = alg_bytes / (head_ms * 1e-3) / 1e9
        out = {
            "metric": "Kalman steps/sec (M outputs x T ticks) + NLL rel-err vs CPU oracle",
            "value": value, "unit": "Kalman steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if dtype == torch.float32 else "f64", "data": "synthetic",
            "config": {"workload": desc, "latents_per_gpu": Lg_per, "latents_total": Lglobal, "ticks": T, "state_dim": bank.d,
                       "layout": ("segment-major [ceil(T/SEG)][L][SEG], SEG = 4 KB of ticks" if tiled else "series-major [L][ld]") + ", HBM-resident", "sharding": f"latents x{world}, NLL scalar all-reduce per pass" + ("" if world == 1 else (" (stream-ordered)" if args.sync_allreduce else " (overlapped with the next pass)"))},
            "roofline": {"bound": "hbm", "achieved": head_achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": head_achieved / HBM_PEAK_GBPS,
                         "frac_is": head_kind,
                         "traffic": traffic, "traffic_source": traffic_source,
                         "kernel": filter_kernel_name(bank, L, T, dtype), "kernel_ms": head_ms,
                         "frac_resident": achieved / HBM_PEAK_GBPS, "achieved_resident": achieved, "kernel_ms_resident": kern_ms,
                         "kernel_ms_from": f"HIP event pairs on every launch of a second loop of {kpasses} passes behind the timed region "
                                           f"({len(kern_samples)} launches; that loop's own wall time: {ms_per_step_bracketed:.4f} ms per pass)",
                         "algorithmic_bytes_per_launch": alg_bytes, "frac_of_measured_copy_peak": head_achieved / HBM_COPY_GBPS,
                         # what a plain copy with this sweep's access pattern reaches on this chip (tools/micro/rows_copy.hip, profiles/r04/
                         # rows_copy_access_pattern.log: 4096 row streams 40 KB apart, one wavefront each): the layout's own ceiling
                         "access_pattern_copy_ceiling_GBps": ({"resident": 6900.0, "cold": 5940.0} if tiled else {"resident": 6820.0, "cold": 5250.0}),
                         "access_pattern_copy_ceiling_source": "profiles/r04/rows_copy_access_pattern.log (one wavefront per latent, 4 KB per step, this layout)",
                         "other_layout": other_layout,
                         # the other wall (mode F is 2 d^2 + 2 d flop per step; the d = 12 fp64 configuration sits on this one)
                         "vector_alu": valu_side(bank.d, dtype, L * min(slab, T) / (kern_ms * 1e-3))},
            "ms_per_step_with_event_pairs": ms_per_step_bracketed,
            "nll_total": float(total.item()),
        }
        if multi and not rehearsal:
            out["collective"] = {"backend": dist.get_backend(), "world": dist.get_world_size(),
                                 "op": "all_reduce(SUM) of the 8-byte NLL total, once per pass" + ("" if args.sync_allreduce else ", overlapped with the next pass")}
        elif multi:
            out["collective"] = {"backend": dist.get_backend(), "world": dist.get_world_size(), "op": "rehearsal"}
        if stub:
            out["rehearsal"] = "BENCH_REHEARSAL=stub: the sweep is a stub on the CPU (launch / rank / reduction logic only): NOT a measurement"
            out["roofline"]["frac"] = None; out["roofline"]["achieved"] = None
        elif rehearsal:
            out["rehearsal"] = "all ranks on one GPU, gloo exchange: numbers are not comparable"
        if force_dist:
            out["forced_dist"] = f"process group built for {world} rank(s) (backend {dist.get_backend()}): every pass ran its all-reduce"
        for k_, v_ in cold.items():
            out["roofline"].setdefault(k_, v_)
        if c4_multi is not None:
            out["other_configs"] = {"c4": c4_multi}
        if world == 1 and not args.no_cpu and not stub:
            if not args.no_others and not force_dist:
                # the other single-GPU rows of BASELINE.json / SURVEY 8, measured in the same run: the other filter shapes (kernel-exact
                # events + wall clock + cold leg), the gradient sweeps (mode G) and configs[2]'s objective evaluation
                others = {}
                for name in sorted(CONFIGS):
                    if name == args.config or name in SLAB:
                        continue
                    others[name] = filter_row(name, device)
                for name in ("c3grad", "c5grad"):
                    g = grad_row(name, 10, 2, cpu=False)
                    others[name] = {"workload": g["config"]["workload"], "state_dim": g["config"]["state_dim"], "dtype": g["dtype"], "ms_per_step": g["ms_per_step"],
                                    "value": g["value"], "kernel": g["roofline"]["kernel"], "kernel_ms": g["roofline"]["kernel_ms"],
                                    "kernel_ms_from": g["roofline"]["kernel_ms_from"], "bound": "valu", "achieved_TFLOPs": g["roofline"]["achieved"],
                                    "peak_TFLOPs": g["roofline"]["peak"], "frac": g["roofline"]["frac"], "flops_per_step": g["roofline"]["flops_per_step"],
                                    "flop_count": g["roofline"]["flop_count"]}
                for nm, dt_ in (("c3e2e", torch.float32), ("c3e2e_f64", torch.float64)):
                    try:
                        others[nm] = e2e_row(device, dt_)
                    except Exception as e:         # (context rows: never let one take the headline down)
                        others[nm] = {"error": str(e)}
                try:
                    others["c4"] = slab_row(device)
                except Exception as e:
                    others["c4"] = {"error": str(e)}
                try:
                    others["c5gaps"] = gaps_row(device)            # configs[4]'s shape with 1 % of the ticks missing, fp64 and fp32
                except Exception as e:
                    others["c5gaps"] = {"error": str(e)}
                try:
                    lr = learn_row(3, 1, cpu=False, windows=(128,))
                    others["c3learn"] = {"workload": lr["config"]["workload"], "dtype": "f64", "ms_per_step": lr["ms_per_step"], "value": lr["value"],
                                         "bound": "mfma", "achieved_TFLOPs": lr["roofline"]["achieved"], "peak_TFLOPs": 78.6, "frac": lr["roofline"]["frac"],
                                         "update_ms": lr["roofline"]["update_ms"], "window_eval_ms": lr["roofline"]["window_eval_ms"],
                                         "host_path_evaluation_ms": lr["roofline"]["host_path_evaluation_ms"], "note": lr["roofline"]["note"]}
                except Exception as e:             # (a context line: never let it take the headline down)
                    others["c3learn"] = {"error": str(e)}
                try:
                    lo = loop_row(ticks=3, threading=True)
                    others["c3loop"] = {"workload": lo["config"]["workload"], "dtype": "f64", "ms_per_step": lo["ms_per_step"], "value": lo["value"],
                                        "objective_evaluations_per_tick": lo["learner"]["objective_evaluations_per_tick"],
                                        "lbfgs_iterations_per_tick": lo["learner"]["lbfgs_iterations_per_tick"],
                                        "ms_per_objective_evaluation_incl_optimiser": lo["ms_per_objective_evaluation_incl_optimiser"]}
                except Exception as e:
                    others["c3loop"] = {"error": str(e)}
                out["other_configs"] = others
            if tiled:                                   # the parity leg reads series-major means
                if nslab > 1:
                    yhat_slabs = [untile_stream(yt_slabs[k], Ty_slabs[k].shape[1]) for k in range(nslab)]
                else:
                    yhat = untile_stream(yt, T)
            if nslab > 1:
                yhat = torch.cat([y[:, :Ty_slabs[k].shape[1]] for k, y in enumerate(yhat_slabs)], dim=1)
            sub = np.arange(0, L, max(1, L // 64))[:64]
            cb, nll_rel, mean_rel = cpu_baseline(prm, Ty[:, :T].cpu().numpy(), T, float(total.item()), yhat[sub][:, :T].double().cpu().numpy(), sub, kernel)
            out["cpu_baseline"] = cb
            out["nll_rel_err"] = nll_rel
            out["filtered_mean_rel_err"] = mean_rel
            out["speedup_vs_cpu_all_cores"] = value / cb["value"]
        print(json.dumps(out), flush=True)
    if multi:
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main() or 0)
