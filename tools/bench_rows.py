#!/usr/bin/env python3
"""Per-row measurements of the hot-path scope table beside the CPU oracle, one JSON object per row.

    python tools/bench_rows.py [--rows update,grad,project,tick,window,stacked] [--L 4096] [--T 10000] [--M 4096]

  update   A7  IHGP::update for L latents (ihgp.h:117-201) on device           vs oracle orc_ihgp_update (all cores)
  grad     A2+A5 sensitivity/gradient sweep over streams (ihgp.h:37-57,:212-222)  vs oracle orc_grad_stream (all cores)
  project  A3  whole-stream OILMM projection / un-projection GEMMs (moihgp.h:181,:222-225)
  tick     A1-A6,A8 behind the per-tick reference ABI: latency per gp32_* call    vs oracle orc_gp_* (1 core)
"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from bench import synth_params, synth_stream, usable_cpus, SEED

FP32_VEC_TF, FP64_VEC_TF = 157.3, 78.6          # MI355X_MICROARCH.md (fp64 vector: AMD public figure)
FP32_MFMA_TF, FP64_MFMA_TF = 157.3, 78.6


def ev_time(fn, n=10, warm=2):
    for _ in range(warm): fn()
    ts = []
    for _ in range(n):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return float(np.median(ts)) * 1e-3


def row_update(a):
    from multioutputihgp_amd.streams import LatentBank
    from oracle import cref
    prm = synth_params(a.L, 0, np.random.default_rng(SEED))
    bank = LatentBank(0.1, prm, kernel="Matern52ss")
    t0 = time.perf_counter(); n = 5
    for _ in range(n): bank.update(prm)
    t_gpu = (time.perf_counter() - t0) / n           # includes H2D of the params and the sync, as a caller sees it
    t0 = time.perf_counter(); cref.ihgp_array("Matern52", 0.1, prm[:512]); t_cpu = (time.perf_counter() - t0) / 512 * a.L
    return dict(row="A7 IHGP::update", latents=a.L, gpu_s=t_gpu, gpu_latents_per_s=a.L / t_gpu, cpu_1core_s=t_cpu,
                cpu_latents_per_s=a.L / t_cpu, note="fp64, one lane per latent; latency-sized, no roofline claimed")


def row_grad(a):
    from multioutputihgp_amd.streams import LatentBank
    from oracle import cref
    out = []
    prm = synth_params(a.L, 0, np.random.default_rng(SEED))
    bank = LatentBank(0.1, prm, kernel="Matern52ss")
    for dtype, name, peak in ((torch.float32, "f32", FP32_VEC_TF), (torch.float64, "f64", FP64_VEC_TF)):
        for T in (128, a.T):
            Ty = synth_stream(a.L, 0, T, dtype, "cuda", SEED + 1)
            x = torch.zeros((a.L, 3), dtype=dtype, device="cuda"); dx = torch.zeros((a.L, 3, 3), dtype=dtype, device="cuda")
            t = ev_time(lambda: bank.grad(Ty, T=T, x=x, dx=dx), n=5, warm=1)
            flops = 221.0 * a.L * T                    # mode G, d=3, P=3
            es = 4 if dtype == torch.float32 else 8
            rec = dict(row="A2+A5 grad sweep", dtype=name, latents=a.L, ticks=T, gpu_s=t, steps_per_s=a.L * T / t,
                       achieved_tflops=flops / t / 1e12, vector_peak_tflops=peak, frac=flops / t / 1e12 / peak,
                       read_GBps=es * a.L * T / t / 1e9, bound="valu")
            if dtype == torch.float64 and T == a.T:
                Ls = min(a.L, 1024)
                igps = cref.ihgp_array("Matern52", 0.1, prm[:Ls])
                Tyh = Ty[:Ls, :T].cpu().numpy()
                nth = max(1, min(usable_cpus(), int(cref.lib().orc_max_threads())))
                t0 = time.perf_counter(); cref.grad_stream(igps, Tyh, want_yhat=False, nthreads=nth); tc = time.perf_counter() - t0
                rec.update(cpu_steps_per_s=Ls * T / tc, cpu_threads=nth)
            out.append(rec)
    return out


def row_project(a):
    from multioutputihgp_amd import MOIHGP
    from multioutputihgp_amd.streams import project_stream, unproject_stream
    out = []
    M = L = a.M
    rng = np.random.default_rng(SEED)
    gp = MOIHGP(0.1, M, L, kernel="Matern52ss")
    for dtype, name, peak in ((torch.float32, "f32", FP32_MFMA_TF), (torch.float64, "f64", FP64_MFMA_TF)):
        T = min(a.T, 4096)
        Y = torch.randn((T, M), dtype=dtype, device="cuda")
        t = ev_time(lambda: project_stream(gp, Y), n=3, warm=1)
        Ty = project_stream(gp, Y)
        t2 = ev_time(lambda: unproject_stream(gp, Ty, T), n=3, warm=1)
        fl = 2.0 * T * M * L
        out.append(dict(row="A3 project_stream", dtype=name, M=M, L=L, T=T, gpu_s=t, achieved_tflops=fl / t / 1e12, mfma_peak_tflops=peak,
                        frac=fl / t / 1e12 / peak, bound="mfma"))
        out.append(dict(row="A3 unproject_stream", dtype=name, M=M, L=L, T=T, gpu_s=t2, achieved_tflops=fl / t2 / 1e12, mfma_peak_tflops=peak,
                        frac=fl / t2 / 1e12 / peak, bound="mfma"))
    return out


def row_stacked(a):
    """Stacked-state latents: update and filter at BASELINE.json's d = 6 / d = 12 shapes, fp64 and fp32,
    beside the CPU port (wide oracle build, all cores)."""
    from multioutputihgp_amd.streams import LatentBank
    from oracle import cref
    out = []
    for kern, L, T in (("Matern52x2", 256, a.T), ("Matern52x2", a.L, a.T), ("Matern52x4", a.L, a.T), ("Matern32x2", a.L, a.T)):
        prm = synth_params(L, 0, np.random.default_rng(SEED), kern)
        bank = LatentBank(0.1, prm, kernel=kern)
        t0 = time.perf_counter()
        for _ in range(5): bank.update(prm)
        t_upd = (time.perf_counter() - t0) / 5
        for dtype, name in ((torch.float64, "f64"), (torch.float32, "f32")):
            Ty = synth_stream(L, 0, T, dtype, "cuda", SEED + 1)
            x = torch.zeros((L, bank.d), dtype=dtype, device="cuda")
            yh = torch.empty_like(Ty); nll = torch.empty((L,), dtype=torch.float64, device="cuda")
            for _ in range(3): bank.filter(Ty, T=T, x=x, yhat=yh, nll=nll)
            bank.profile_enable(20)
            for _ in range(20):
                x.zero_(); bank.filter(Ty, T=T, x=x, yhat=yh, nll=nll)
            ms = float(np.mean(bank.profile_read()))
            es = 4 if dtype == torch.float32 else 8
            rec = dict(row="stacked filter+NLL", kernel=kern, state_dim=bank.d, dtype=name, latents=L, ticks=T, update_ms=t_upd * 1e3,
                       kernel_ms=ms, steps_per_s=L * T / (ms * 1e-3), algorithmic_GBps=2 * es * L * T / (ms * 1e-3) / 1e9,
                       frac_hbm=2 * es * L * T / (ms * 1e-3) / 1e9 / 8000.0)
            if dtype == torch.float64 and L >= 1024:
                Ls = 512
                nth = max(1, min(usable_cpus(), int(cref.lib(wide=True).orc_max_threads())))
                igps = cref.ihgp_array(kern, 0.1, prm[:Ls])
                Tyh = Ty[:Ls, :T].cpu().numpy()
                t0 = time.perf_counter(); cref.filter_stream(igps, Tyh, nthreads=nth); tc = time.perf_counter() - t0
                rec.update(cpu_steps_per_s=Ls * T / tc, cpu_threads=nth, cpu_sample=f"{Ls} latents x {T} ticks, wide oracle build, OpenMP")
            out.append(rec)
    return out


def row_tick(a):
    from multioutputihgp_amd import MOIHGP
    from oracle import cref
    out = []
    rng = np.random.default_rng(SEED)
    for (M, L) in ((8, 4), (64, 64), (256, 256), (1024, 1024)):
        gp = MOIHGP(0.1, M, L, kernel="Matern32"); ref = cref.GP(0.1, M, L, "Matern32"); ref.set_literal_ugrad(0)
        p = gp.params.copy(); ref.update(p); gp.update(p)
        x = rng.standard_normal((L, 2)); dx = rng.standard_normal((L, 3, 2)); y = rng.standard_normal(M)
        def tm(f, n):
            f(); t0 = time.perf_counter()
            for _ in range(n): f()
            return (time.perf_counter() - t0) / n
        n = 50 if L <= 256 else 10
        rec = dict(row="per-tick reference ABI", M=M, L=L,
                   gpu_step3_us=tm(lambda: gp.step(x, y), n) * 1e6, gpu_step1_us=tm(lambda: gp.step(x, y, dx), n) * 1e6,
                   gpu_lik1_us=tm(lambda: gp.negLogLikelihood(x, y, dx), n) * 1e6,
                   cpu_step3_us=tm(lambda: ref.step(x, y), n) * 1e6, cpu_step1_us=tm(lambda: ref.step(x, y, dx), n) * 1e6,
                   cpu_lik1_us=tm(lambda: ref.negLogLikelihood(x, y, dx), n) * 1e6,
                   note="wall time per call incl. H2D/D2H copies and sync; CPU = oracle with O(M L) projection and closed-form U-gradient, 1 core")
        t0 = time.perf_counter(); gp.update(p); rec["gpu_update_ms"] = (time.perf_counter() - t0) * 1e3
        t0 = time.perf_counter(); ref.update(p); rec["cpu_update_ms"] = (time.perf_counter() - t0) * 1e3
        out.append(rec)
    return out


def row_window(a):
    """Config 3's objective evaluation: update(params) + the windowed NLL/gradient sweep, as the learner calls it
    (moihgp_online.h:40-72)."""
    from multioutputihgp_amd import MOIHGP
    out = []
    rng = np.random.default_rng(SEED)
    for (M, L) in ((256, 256), (a.M, a.M)):
        gp = MOIHGP(0.1, M, L, kernel="Matern52ss")
        p = gp.params.copy()
        dU = rng.standard_normal(M * L)
        p[:M * L] += 0.1 * dU / np.linalg.norm(dU)          # ||dU||_F = 0.1 = the reference's L-BFGS-B max_step (moihgp_online.h:156)
        p[M * L + L + 1:] = synth_params(L, 0, rng).ravel()
        d = gp.igp_dim
        x = np.zeros((L, d)); dx = np.zeros((L, 3, d))
        for W in (16, 128):
            Y = 0.5 * rng.standard_normal((W, M))
            gp.update(p); gp.window_objective(Y, x, dx)
            n = 5
            t0 = time.perf_counter()
            for _ in range(n): gp.update(p)
            t_upd = (time.perf_counter() - t0) / n
            gp.window_objective(Y, x, dx)
            t0 = time.perf_counter()
            for _ in range(n): gp.window_objective(Y, x, dx, set_window=False)
            t_ev = (time.perf_counter() - t0) / n
            flops = 3 * 2.0 * W * M * L + 221.0 * L * W
            out.append(dict(row="N2 window objective", M=M, L=L, W=W, update_ms=t_upd * 1e3, eval_ms=t_ev * 1e3,
                            eval_includes="H2D of x/dx, 3 GEMMs, sweep, reductions, D2H of loss + grad[M*L+L+1+3L]",
                            grad_entries=gp.num_param, approx_tflops=flops / t_ev / 1e12))
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", default="update,grad,project,tick,window,stacked")
    ap.add_argument("--L", type=int, default=4096); ap.add_argument("--T", type=int, default=10000); ap.add_argument("--M", type=int, default=4096)
    a = ap.parse_args()
    torch.cuda.set_device(0)
    for r in a.rows.split(","):
        res = {"update": row_update, "grad": row_grad, "project": row_project, "tick": row_tick, "window": row_window, "stacked": row_stacked}[r](a)
        for rec in (res if isinstance(res, list) else [res]):
            print(json.dumps(rec), flush=True)
