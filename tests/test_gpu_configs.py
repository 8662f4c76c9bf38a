"""GPU suite (-m gpu): BASELINE.json's configurations that the kernel-level tests do not reach at their worded sizes.

C1  configs[0]  "example_regression: M=4 outputs, T=500, Matern-3/2, fp64": 500 ticks through the per-tick reference ABI
                (gp32_step3, one FFI crossing per tick, exactly as example.py:40-42 drives it) against the oracle's tick loop;
                plus BASELINE's "d=4" reading of it: a bank of stacked 2 x Matern-3/2 latents over the same projected stream.
C4  configs[3]  "M=32768, T=100000 streamed, fp32, sharded 8 x MI355X": ONE GPU's shard, 4096 latents x 10^5 ticks in ten 10^4-tick slabs
                that carry the state: a 64-latent subset against the oracle, slab carry == one launch, additivity of the NLL over
                slabs; and two ranks (gloo, both on the one GPU of the test box) whose 2048-latent shards add up to the 4096-latent
                total through the path's only exchange, the 8-byte all-reduce.
Tolerances (BASELINE.json north_star): 1e-6 relative fp64, 1e-3 fp32 on filtered means and NLL; tighter where the arithmetic allows."""
import numpy as np
import pytest
import torch

from conftest import rel_err, rel_err_rows

pytestmark = pytest.mark.gpu
SEED = 20260101


@pytest.fixture(scope="module")
def env(hip_built):
    assert torch.cuda.is_available(), "GPU suite needs a GPU"
    torch.cuda.set_device(0)
    from multioutputihgp_amd import MOIHGP, load_library
    from multioutputihgp_amd import streams
    from oracle import cref
    assert load_library().moihgp_device_count() >= 1
    return dict(MOIHGP=MOIHGP, streams=streams, cref=cref)


def synth_params(L, rng):
    return np.column_stack([rng.uniform(0.5, 2, L), rng.uniform(0.5, 2, L), rng.uniform(0.05, 0.2, L)])


# ------------------------------------------------------------------------------------------ C1
def test_c1_example_shape_through_the_per_tick_abi(env):
    """configs[0]: M = L = 4, T = 500, Matern-3/2, fp64, one gp32_step3 per tick (example.py:40-42), then the same ticks through
    step1 (with sensitivities) + lik1 / lik2 as the learners call them (online_learning.py:84-89)."""
    M = L = 4; T = 500
    rng = np.random.default_rng(SEED)
    gp = env["MOIHGP"](0.1, M, L, kernel="Matern32")
    ref = env["cref"].GP(0.1, M, L, "Matern32"); ref.set_literal_ugrad(0)
    params = np.concatenate([(np.eye(M, L) + 0.1 * rng.standard_normal((M, L))).ravel(), rng.uniform(0.5, 2, L), [0.05], synth_params(L, rng).ravel()])
    gp.update(params); ref.update(params)
    Y = np.sin(0.05 * np.arange(T)[:, None] * (1 + np.arange(M)[None, :])) + 0.1 * rng.standard_normal((T, M))
    x = np.zeros((L, 2)); xr = np.zeros((L, 2))
    out, out_r = np.empty((T, M)), np.empty((T, M))
    for t in range(T):
        x, out[t] = gp.step(x, Y[t])
        xr, out_r[t] = ref.step(xr, Y[t])
    assert rel_err(out, out_r) < 1e-9 and rel_err(x, xr) < 1e-9
    # the learners' loop on the same stream: step with sensitivities, NLL on the pre-step state (moihgp_online.h:64-66)
    x, dx = np.zeros((L, 2)), np.zeros((L, 3, 2)); xr, dxr = x.copy(), dx.copy()
    loss, loss_r, grad, grad_r = 0.0, 0.0, np.zeros(gp.num_param), np.zeros(gp.num_param)
    for t in range(60):
        l1, g1 = gp.negLogLikelihood(x, Y[t], dx); l2, g2 = ref.negLogLikelihood(xr, Y[t], dxr)
        loss += l1; loss_r += l2; grad += g1; grad_r += g2
        assert abs(gp.negLogLikelihood(x, Y[t]) - ref.negLogLikelihood(xr, Y[t])) < 1e-9 * abs(ref.negLogLikelihood(xr, Y[t]))
        x, _, dx = gp.step(x, Y[t], dx); xr, _, dxr = ref.step(xr, Y[t], dxr)
    assert abs(loss - loss_r) < 1e-9 * abs(loss_r) and rel_err(grad, grad_r) < 1e-8
    # the batched entries on the same 500 ticks: project -> sweep -> unproject == the tick loop
    from multioutputihgp_amd.streams import LatentBank, project_stream, unproject_stream
    Yd = torch.from_numpy(Y).cuda()
    Ty = project_stream(gp, Yd)
    yl, xT, nll = LatentBank.from_handle(gp).filter(Ty, T=T)
    Yhat = unproject_stream(gp, yl, T)
    torch.cuda.synchronize()
    assert rel_err(Yhat.cpu().numpy(), out_r) < 1e-9


def test_c1_d4_reading_stacked_matern32_bank(env):
    """BASELINE.json words configs[0] "Matern-3/2 d=4": the reference's Matern-3/2 has d = 2 (matern32ss.h:95); d = 4 is the stacked
    2 x Matern-3/2 latent.  Four such latents over 500 ticks of the projected C1 stream, against the oracle."""
    L, T = 4, 500
    rng = np.random.default_rng(SEED + 1)
    prm = np.column_stack([rng.uniform(0.5, 2, L), rng.uniform(0.5, 2, L), rng.uniform(0.5, 2, L), rng.uniform(0.5, 2, L), rng.uniform(0.05, 0.2, L)])
    Ty = np.sin(0.05 * np.arange(T)[None, :] * (1 + np.arange(L)[:, None])) + 0.1 * rng.standard_normal((L, T))
    bank = env["streams"].LatentBank(0.1, prm, kernel="Matern32x2")
    assert bank.d == 4
    o = env["cref"].filter_stream(env["cref"].ihgp_array("Matern32x2", 0.1, prm), Ty)
    from multioutputihgp_amd.streams import alloc_stream
    Tyd = alloc_stream(L, T, torch.float64); Tyd.zero_(); Tyd[:, :T] = torch.from_numpy(Ty)
    yhat, xT, nll = bank.filter(Tyd, T=T)
    torch.cuda.synchronize()
    ok = np.isfinite(o["yhat"]).all(axis=1) & (np.abs(o["yhat"]).max(axis=1) < 1e6)       # (the literal DARE can give an unstable stacked latent)
    assert ok.sum() >= 2
    assert rel_err(yhat[:, :T].cpu().numpy()[ok], o["yhat"][ok]) < 1e-9 and rel_err(nll.cpu().numpy()[ok], o["nll_per_latent"][ok]) < 1e-9


# ------------------------------------------------------------------------------------------ C4
def _c4_inputs(L, T, lo=0):
    """bench.py's C4 shard inputs (same generator, same seed): fp32 stream, 4096 latents per GPU."""
    import bench
    prm = bench.synth_params(4096, 0, np.random.default_rng(bench.SEED))[lo:lo + L]
    Ty = bench.synth_stream(4096, 0, T, torch.float32, torch.device("cuda", 0), bench.SEED + 1)[lo:lo + L]
    return prm, Ty


def test_c4_shard_in_slabs(env):
    """One GPU's shard of configs[3]: 4096 latents x 10^5 ticks, fp32, swept as ten 10^4-tick slabs that carry the state."""
    L, T, SLAB = 4096, 100000, 10000
    prm, Ty = _c4_inputs(L, T)
    bank = env["streams"].LatentBank(0.1, prm, kernel="Matern52ss")
    # (a) the whole stream in ONE launch
    y1, x1, n1 = bank.filter(Ty, T=T)
    # (b) ten slabs, each its own contiguous buffer (as slabs of a stream arrive), the state carried from launch to launch
    x = torch.zeros((L, 3), dtype=torch.float32, device="cuda")
    nll_slabs, y_slabs = [], []
    for k in range(T // SLAB):
        slab = Ty[:, k * SLAB:(k + 1) * SLAB].contiguous()
        yk, x, nk = bank.filter(slab, T=SLAB, x=x)
        y_slabs.append(yk[:, :SLAB]); nll_slabs.append(nk.clone())
    torch.cuda.synchronize()
    y2 = torch.cat(y_slabs, dim=1)
    n2 = torch.stack(nll_slabs).sum(dim=0)
    # slab carry == one launch (fp32 rounding order differs at the slab boundaries only: 10^4 is not a multiple of the 1024-tick segment)
    assert rel_err_rows(y2.double().cpu().numpy(), y1[:, :T].double().cpu().numpy()) < 2e-5
    assert rel_err_rows(x.double().cpu().numpy(), x1.double().cpu().numpy(), floor=1e-3) < 1e-4
    # additivity of the NLL over slabs: per latent and in total
    assert rel_err(n2.cpu().numpy(), n1.cpu().numpy()) < 1e-6
    assert abs(n2.sum().item() - n1.sum().item()) < 1e-7 * abs(n1.sum().item())
    # a 64-latent subset against the fp64 oracle on identical inputs (the bar: 1e-3 on filtered means and NLL)
    sub = np.arange(0, L, L // 64)[:64]
    o = env["cref"].filter_stream(env["cref"].ihgp_array("Matern52", 0.1, prm[sub]), Ty[sub][:, :T].double().cpu().numpy(), nthreads=8)
    assert rel_err_rows(y2[sub].double().cpu().numpy(), o["yhat"]) < 1e-3
    assert rel_err(n2[sub].cpu().numpy(), o["nll_per_latent"]) < 1e-4
    # and the oracle's own fp32 loop (same arithmetic type) on a few of them, tighter
    of = env["cref"].filter_stream(env["cref"].ihgp_array("Matern52", 0.1, prm[sub[:8]]), Ty[sub[:8]][:, :T].cpu().numpy(), nthreads=8)
    assert rel_err_rows(y2[sub[:8]].cpu().numpy(), of["yhat"]) < 1e-4


def _c4_rank(rank, world, port, q):
    import os
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)       # 2 ranks share the one GPU of the test box: exchange on gloo
    torch.cuda.set_device(0)
    from multioutputihgp_amd.sharded import allreduce_nll, shard_bounds
    from multioutputihgp_amd.streams import LatentBank
    Lg, T, SLAB = 4096, 20000, 10000
    lo, hi = shard_bounds(Lg, world, rank)
    prm, Ty = _c4_inputs(hi - lo, T, lo)
    bank = LatentBank(0.1, prm, kernel="Matern52ss")
    x = torch.zeros((hi - lo, 3), dtype=torch.float32, device="cuda")
    acc = torch.zeros((hi - lo,), dtype=torch.float64, device="cuda")
    for k in range(T // SLAB):
        _, x, nk = bank.filter(Ty[:, k * SLAB:(k + 1) * SLAB].contiguous(), T=SLAB, x=x, want_yhat=False)
        acc += nk
    total = allreduce_nll(acc)                                          # the path's only exchange: 8 bytes
    torch.cuda.synchronize()
    q.put((rank, lo, hi, float(total.item()), float(acc.sum().item())))
    dist.destroy_process_group()


def test_c4_two_shards_add_up_to_the_unsharded_total(env):
    """configs[3] shards the latents over ranks with no data-path collective; the only exchange is the all-reduce of the NLL scalar.
    Two 2048-latent shards (two processes, gloo) against one 4096-latent bank on the same streams."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_c4_rank, args=(r, 2, port, q)) for r in range(2)]
    for p in procs: p.start()
    res = sorted([q.get(timeout=300) for _ in procs], key=lambda r: r[0])
    for p in procs: p.join(timeout=60)
    (_, lo0, hi0, tot0, own0), (_, lo1, hi1, tot1, own1) = res
    assert (lo0, hi0, lo1, hi1) == (0, 2048, 2048, 4096)
    assert tot0 == tot1 and abs(tot0 - (own0 + own1)) <= 1e-12 * abs(tot0)
    prm, Ty = _c4_inputs(4096, 20000)
    bank = env["streams"].LatentBank(0.1, prm, kernel="Matern52ss")
    _, _, nll = bank.filter(Ty, T=20000, want_yhat=False)
    torch.cuda.synchronize()
    assert abs(nll.sum().item() - tot0) < 1e-6 * abs(tot0)             # (one 2 x 10^4 launch vs two 10^4 slabs: fp32 rounding at the slab seam)


# ------------------------------------------------------------------------------------------ gradient sweeps at BASELINE's sizes
@pytest.mark.parametrize("kern,dtype", [("Matern52x4", torch.float64), ("Matern52x2", torch.float64), ("Matern52", torch.float32)])
def test_gradient_sweep_at_full_size(env, kern, dtype):
    """Mode G at the sizes of configs[4] / [1] / [2]: 4096 latents x 10^4 ticks, d = 12 / 6 / 3.  Checks that do not need
    a full-size oracle run: (1) a 16-latent subset against the oracle (NLL, gradient, final x and dx), (2) a sweep cut in two
    that carries (x, dx) -- at a tick that is neither a segment nor a chunk boundary -- gives the same final state and, added up,
    the same NLL and gradient as one sweep (ihgp.h:37-57 is a recursion in (x, dx); :215-219 a sum over ticks)."""
    import bench
    L, T = 4096, 10000
    rng = np.random.default_rng(23)
    stacked = "x" in kern
    prm = bench.synth_params(L, 0, rng, kern if stacked else "Matern52ss")
    bank = env["streams"].LatentBank(0.1, prm, kernel=kern if stacked else "Matern52ss")
    Ty = rng.standard_normal((L, T))
    Tyd = torch.from_numpy(Ty).to(dtype).cuda()
    tol = 1e-9 if dtype == torch.float64 else 2e-4
    r = bank.grad(Tyd, T=T, want_yhat=False)
    torch.cuda.synchronize()
    sub = np.sort(rng.choice(L, size=16, replace=False))
    o = env["cref"].grad_stream(env["cref"].ihgp_array(kern, 0.1, prm[sub]), np.ascontiguousarray(Ty[sub]), want_yhat=False, nthreads=4)
    def err(a, b):
        return float(np.abs(a - b).max() / np.abs(b).max())
    e = (err(r["nll"][sub].cpu().numpy(), o["nll_per_latent"]), err(r["grad"][sub].cpu().numpy(), o["grad"]),
         err(r["x"][sub].cpu().numpy(), o["x"]), err(r["dx"][sub].double().cpu().numpy(), o["dx"]))
    print(f"gradient sweep {kern} {dtype}: subset vs oracle nll {e[0]:.2e} grad {e[1]:.2e} x {e[2]:.2e} dx {e[3]:.2e}")
    assert max(e) < tol * 10, e
    cut = 2 * 2048 + 32 * 7 + 4                                        # (rows of both parts stay 16-byte multiples)
    a = bank.grad(Tyd[:, :cut].contiguous(), T=cut, want_yhat=False)
    b = bank.grad(Tyd[:, cut:].contiguous(), T=T - cut, x=a["x"].clone(), dx=a["dx"].clone(), want_yhat=False)
    torch.cuda.synchronize()
    d = (err((a["nll"] + b["nll"]).cpu().numpy(), r["nll"].cpu().numpy()), err((a["grad"] + b["grad"]).cpu().numpy(), r["grad"].cpu().numpy()),
         err(b["x"].cpu().numpy(), r["x"].cpu().numpy()), err(b["dx"].double().cpu().numpy(), r["dx"].double().cpu().numpy()))
    assert max(d) < tol, d


def test_dev_entries_take_the_callers_stream(env):
    """moihgp_update_dev_on / moihgp_window_eval_dev_on (include/moihgp.h): the operands may still be in flight on the caller's stream when
    the call is made, and the results are ordered on that stream -- no host synchronisation in between.  200 iterations: a producer
    (a large GEMM to keep the stream busy, then the arithmetic that writes the parameter vector and the start state) is queued on a torch
    SIDE stream and the two entries are called at once; the results are copied out on the same stream.  Bit-equal to the path that
    synchronises before every call and uses the plain entries."""
    import ctypes as C
    from multioutputihgp_amd import load_library
    lib = load_library()
    M, L, W, d, P = 96, 48, 8, 3, 3
    rng = np.random.default_rng(SEED + 11)
    dev = torch.device("cuda", 0)
    gp = env["MOIHGP"](0.1, M, L, kernel="Matern52ss")
    S = rng.uniform(0.5, 2.0, L); sigma = 0.04
    base = np.concatenate([(np.eye(M, L) + 0.02 * rng.standard_normal((M, L))).ravel(), S, [sigma], synth_params(L, rng).ravel()])
    delta = np.concatenate([0.01 * rng.standard_normal(M * L), np.zeros(L + 1 + L * P)])
    gp.set_window(0.5 * rng.standard_normal((W, M)))
    base_d, delta_d = torch.from_numpy(base).to(dev), torch.from_numpy(delta).to(dev)
    x0_d = torch.from_numpy(0.2 * rng.standard_normal((L, d))).to(dev); dx0_d = torch.from_numpy(0.05 * rng.standard_normal((L, P, d))).to(dev)
    busy = torch.randn((2048, 2048), device=dev)
    n_it = 200
    def run(synced):
        losses = torch.zeros(n_it, dtype=torch.float64, device=dev); gsum = torch.zeros(n_it, dtype=torch.float64, device=dev)
        gfirst = None
        side = torch.cuda.Stream()
        p_d = torch.empty_like(base_d); x_d = torch.empty_like(x0_d); g_d = torch.empty_like(base_d); l_d = torch.zeros(1, dtype=torch.float64, device=dev)
        torch.cuda.synchronize()
        for k in range(n_it):
            with torch.cuda.stream(side):
                if not synced:
                    (busy @ busy).sum()                                   # keeps the stream busy in front of the producer
                torch.add(base_d, delta_d, alpha=float(k % 7), out=p_d)   # the producer: parameters and start state of this iteration
                torch.mul(x0_d, 1.0 + 0.01 * (k % 5), out=x_d)
                if synced:
                    side.synchronize()
                    rc = lib.moihgp_update_dev(gp.handle, C.c_void_p(p_d.data_ptr()))
                    assert rc == 0
                    rc = lib.moihgp_window_eval_dev(gp.handle, C.c_void_p(x_d.data_ptr()), C.c_void_p(dx0_d.data_ptr()), C.c_void_p(l_d.data_ptr()),
                                                    C.c_void_p(g_d.data_ptr()), None, None)
                    assert rc == 0
                else:
                    gp.update_dev(p_d, stream=side)
                    gp.window_objective_dev(x_d, dx0_d, l_d, g_d, stream=side)
                losses[k:k + 1].copy_(l_d)                                # consumers on the same stream
                gsum[k:k + 1].copy_(g_d.sum().reshape(1))
                if k == n_it - 1:
                    gfirst = g_d.clone()
        side.synchronize()
        return losses.cpu().numpy(), gsum.cpu().numpy(), gfirst.cpu().numpy()
    a = run(True)
    b = run(False)
    assert np.isfinite(a[0]).all() and len(set(a[0].tolist())) > 5
    for u, v in zip(a, b):
        assert np.array_equal(u, v)


# ------------------------------------------------------------------------------------------ configs[2] at its worded size
def test_c3_learning_loop_at_full_size(env):
    """BASELINE.json configs[2] as worded -- "M=4096 outputs, T=10000, Matern-5/2, fp32, online-learning L-BFGS outer loop" -- at
    M = L = 4096: what the optimiser calls per line-search point (moihgp_online.h:40-72) is MOIHGP::update (moihgp.h:431-457: polar
    factor of the 4096 x 4096 mixing + IHGP::update of every latent) and the window loop of step + negLogLikelihood with gradient
    (16.8 M entries); then the filter over the real 10^4-tick observation stream (project -> sweep in fp32 -> unproject).
      (1) update:  ||U^T U - I||  and  U against the LAPACK-SVD polar factor svdU svdV^T (moihgp.h:438-446);
      (2) moihgp_window_eval over W = 2 ticks == this handle's own gp52_lik1 / gp52_step1 tick by tick, both `threading` values;
      (3) W = 16: per-latent gradient block, per-latent losses and carried state of ALL latents against the oracle's gradient sweep
          over the host-projected window; sampled grad_U[r, c] against the closed form evaluated on the host; the device-resident
          entries (moihgp_update_dev / moihgp_window_eval_dev) against the host forms;
      (4) project_stream -> filter (fp32) -> unproject_stream on Y [10^4][4096] against the oracle's per-latent loop on the
          host-projected stream, compared on 16 outputs (1e-3: north_star's fp32 bar)."""
    import ctypes as C
    import scipy.linalg
    from multioutputihgp_amd import load_library
    from multioutputihgp_amd.streams import LatentBank, project_stream, unproject_stream
    cref = env["cref"]
    lib = load_library()
    M = L = 4096; d = 3; P = 3
    rng = np.random.default_rng(SEED + 2)
    gp = env["MOIHGP"](0.1, M, L, kernel="Matern52ss")
    A0 = np.eye(M, L) + 0.004 * rng.standard_normal((M, L))            # singular values in about [0.5, 1.5]: the scaled Newton-Schulz start
    S = rng.uniform(0.5, 2.0, L); sigma = 0.04
    igp = synth_params(L, rng)
    p = np.concatenate([A0.ravel(), S, [sigma], igp.ravel()])
    gp.update(p)
    # ---- (1) the polar factor
    U = gp.params[:M * L].reshape(M, L).copy()
    G = U.T @ U - np.eye(L)
    assert np.abs(G).max() < 1e-12 and np.abs(G).sum(axis=1).max() < 1e-10, (np.abs(G).max(), np.abs(G).sum(axis=1).max())
    u, _, vt = scipy.linalg.svd(A0, full_matrices=False, lapack_driver="gesdd")
    assert np.abs(U - u @ vt).max() < 1e-9
    del u, vt, G
    assert np.array_equal(gp.params[M * L:M * L + L], S) and gp.params[M * L + L] == sigma
    # the learner's case: previous factor + an L-BFGS step of norm 0.1 (moihgp_online.h:156): the unscaled start, fewer steps
    dU = rng.standard_normal((M, L)); p2 = p.copy(); p2[:M * L] = (U + 0.1 * dU / np.linalg.norm(dU)).ravel()
    its_general = lib.moihgp_polar_iterations(gp.handle)
    gp.update(p2)
    assert 0 < lib.moihgp_polar_iterations(gp.handle) < its_general
    U = gp.params[:M * L].reshape(M, L).copy()
    assert np.abs(U.T @ U - np.eye(L)).max() < 1e-12
    # ---- (2) W = 2 window == two ticks of the per-tick reference ABI on the same handle
    igps = cref.ihgp_array("Matern52", 0.1, igp)
    x0 = 0.2 * rng.standard_normal((L, d)); dx0 = 0.05 * rng.standard_normal((L, P, d))
    Y2 = 0.5 * rng.standard_normal((2, M))
    for threading in (0, 1):
        lib.moihgp_set_threading(gp.handle, threading)
        loss_w, grad_w, xw, dxw = gp.window_objective(Y2, x0, dx0)
        grad_w = grad_w.copy()
        x, dx, loss_t, grad_t = x0, dx0, 0.0, np.zeros(gp.num_param)
        for t in range(2):
            l1, g1 = gp.negLogLikelihood(x, Y2[t], dx)
            xn, _, dxn = gp.step(x, Y2[t], dx)
            loss_t += l1; grad_t += g1; x, dx = xn, dxn
        assert abs(loss_w - loss_t) < 1e-9 * abs(loss_t), (threading, loss_w, loss_t)
        assert rel_err(grad_w, grad_t) < 1e-9 and rel_err(grad_w[M * L:], grad_t[M * L:]) < 1e-9
        assert rel_err(xw, x) < 1e-12 and rel_err(dxw, dx) < 1e-11
        if threading == 0:
            loss_off = loss_w
        else:                                                          # moihgp.h:590: the threaded branch adds the per-latent losses
            Ty2 = (U.T @ Y2.T) / np.sqrt(S)[:, None]
            o2 = cref.grad_stream(igps, Ty2, x0=x0, dx0=dx0, want_yhat=False, nthreads=8)
            assert abs((loss_w - loss_off) - o2["nll"]) < 1e-9 * abs(o2["nll"])
    lib.moihgp_set_threading(gp.handle, 0)
    del grad_t, g1
    # ---- (3) W = 16 against the oracle, all latents
    W = 16
    Y = 0.5 * rng.standard_normal((W, M))
    loss, grad, xT, dxT = gp.window_objective(Y, x0, dx0)
    grad = grad.copy()
    Ty = (U.T @ Y.T) / np.sqrt(S)[:, None]                             # moihgp.h:181 on the host
    o = cref.grad_stream(igps, Ty, x0=x0, dx0=dx0, want_yhat=False, nthreads=8)
    gl = grad[M * L + L + 1:].reshape(L, P)
    assert rel_err(gl, o["grad"]) < 1e-9                               # moihgp.h:608-609: the per-latent blocks
    assert rel_err(xT, o["x"]) < 1e-9 and rel_err(dxT, o["dx"]) < 1e-8
    # global terms of the loss (moihgp.h:503, threading off: no per-latent losses) and the sigma / S entries (moihgp.h:555-563, :604-605)
    UUY = (U @ (U.T @ Y.T)).T
    rt = np.linalg.norm(Y - UUY, axis=1)
    m_n = max(M - L, 0)
    assert abs(loss - (W * (0.5 * np.log(S.sum()) + 0.5 * m_n * np.log(sigma)) + 0.5 * rt.sum() / sigma)) < 1e-9 * abs(loss)
    # closed-form U-gradient  grad_U[r, c] = sum_t y_t[r] (pv_{c,t} / sqrt(S_c) - (U^T y_t)_c / sigma),  pv = (y_t[c] - HA x_{c,t}) (1 - HA K) / S_igp
    cols = np.sort(rng.choice(L, size=12, replace=False)); rows = np.sort(rng.choice(M, size=12, replace=False))
    Uty = U.T @ Y.T                                                    # [L][W]
    gU = grad[:M * L].reshape(M, L)
    spu = np.zeros(L)
    for c in cols:
        Ai, Ki, HAi, Si = igps[c].mat("A"), igps[c].mat("K"), igps[c].mat("HA"), float(igps[c].mat("S"))
        xc = x0[c].copy(); z = np.empty(W)
        for t in range(W):
            hx = HAi @ xc
            pv = (Y[t, c] - hx) * (1 - HAi @ Ki) / Si                  # moihgp.h:510-511 (raw y(c), sic)
            z[t] = pv / np.sqrt(S[c]) - Uty[c, t] / sigma
            spu[c] += pv * Uty[c, t]
            xc = Ai @ xc + Ki * (Ty[c, t] - hx)
        want = Y[:, rows].T @ z
        assert rel_err(gU[rows, c], want) < 1e-9, c
        gS = W * 0.5 / S[c] - 0.5 * S[c] ** -1.5 * spu[c] - gl[c, 2] * sigma / S[c] ** 2          # moihgp.h:555-561, :604
        assert abs(grad[M * L + c] - gS) < 1e-9 * max(1.0, abs(gS)), c
    g_sigma = 0.5 * (W * m_n - rt.sum() / sigma) / sigma + (gl[:, 2] / S).sum()                  # moihgp.h:563, :605
    assert abs(grad[M * L + L] - g_sigma) < 1e-9 * abs(g_sigma)
    # ---- the device-resident entries: same kernels, same values
    dev = torch.device("cuda", 0)
    p_dev = torch.from_numpy(p2).to(dev); g_dev = torch.empty_like(p_dev); l_dev = torch.zeros(1, dtype=torch.float64, device=dev)
    xn_dev = torch.empty((L, d), dtype=torch.float64, device=dev); dxn_dev = torch.empty((L, P, d), dtype=torch.float64, device=dev)
    gp.update_dev(p_dev)
    assert lib.moihgp_polar_iterations(gp.handle) > 0
    gp.window_objective_dev(torch.from_numpy(x0).to(dev), torch.from_numpy(dx0).to(dev), l_dev, g_dev, xn_dev, dxn_dev)
    assert l_dev.item() == loss and torch.equal(g_dev.cpu(), torch.from_numpy(grad))
    assert np.array_equal(xn_dev.cpu().numpy(), xT) and np.array_equal(dxn_dev.cpu().numpy(), dxT)
    back = torch.empty_like(p_dev)
    gp.params_dev(back)
    assert np.array_equal(back.cpu().numpy(), gp.params)
    del p_dev, g_dev, back, grad, gU
    # ---- (4) the filter over the real stream: Y [10^4][4096] fp32
    T = 10000
    t_ax = np.arange(T)[:, None]
    Yr = (np.sin(0.05 * t_ax * (1 + np.arange(M)[None, :] % 7)) + 0.1 * rng.standard_normal((T, M))).astype(np.float32)
    Yd = torch.from_numpy(Yr).cuda()
    Tyd = project_stream(gp, Yd)
    yl, xe, nll = LatentBank.from_handle(gp).filter(Tyd, T=T)
    Yhat = unproject_stream(gp, yl, T)
    torch.cuda.synchronize()
    Ty_h = (U.T @ Yr.astype(np.float64).T) / np.sqrt(S)[:, None]      # [L][T] fp64 on the host
    assert rel_err_rows(Tyd[:, :T].double().cpu().numpy(), Ty_h) < 1e-3
    of = cref.filter_stream(igps, Ty_h, nthreads=16)
    outs = np.sort(rng.choice(M, size=16, replace=False))
    want = (U[outs] * np.sqrt(S)[None, :]) @ of["yhat"]                # moihgp.h:222-225 for 16 outputs: [16][T]
    got = Yhat[:, outs].double().cpu().numpy().T
    e_mean = rel_err_rows(got, want)
    e_nll = abs(nll.sum().item() - of["nll"]) / abs(of["nll"])
    print(f"configs[2] filter at M=L=4096, T=1e4, fp32: filtered means (16 outputs) {e_mean:.2e}, NLL {e_nll:.2e}")
    assert e_mean < 1e-3 and e_nll < 1e-3


# ------------------------------------------------------------------------------------------ RCCL-only code on the one GPU of the test box
def _clean_env(**extra):
    import os
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "BENCH_REHEARSAL", "BENCH_FORCE_DIST"):
        env.pop(k, None)
    env.update(extra)
    return env


def _free_port():
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return str(p)


def test_rccl_only_code_on_a_one_rank_communicator(env):
    """No multi-GPU box is available to the build: the branches only an RCCL communicator takes (async all-reduce on its stream,
    reduce_scatter_tensor with padding, all_gather, init with device_id) run here on a ONE-rank communicator, the world == 1
    short-cuts lifted (sharded.FORCE_COLLECTIVES); in a process of its own (tests/rccl_one_rank.py)."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "rccl_one_rank.py")], capture_output=True, text=True, timeout=600,
                       env=_clean_env(MASTER_ADDR="127.0.0.1", MASTER_PORT=_free_port()))
    assert r.returncode == 0 and r.stdout.strip().endswith("OK"), (r.stdout[-2000:], r.stderr[-4000:])


def test_bench_multi_gpu_code_on_a_one_rank_communicator(env):
    """bench.py's N > 1 code (process group on the RCCL backend, every pass ending in its all-reduce, overlapped with the next sweep,
    max over ranks on the device) on one rank: BENCH_FORCE_DIST=1."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "10", "--warmup", "2", "--no-cpu", "--no-cold"],
                       capture_output=True, text=True, timeout=600,
                       env=_clean_env(BENCH_FORCE_DIST="1", WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=_free_port()))
    assert r.returncode == 0, r.stderr[-4000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["n_gpus"] == 1 and "nccl" in out["forced_dist"] and out["value"] > 1e11 and np.isfinite(out["nll_total"])


def test_bench_rehearsal_two_ranks_on_one_gpu(env):
    """`python bench.py --gpus 2` exactly as the driver runs it (no launcher): the parent starts the ranks itself.  Both ranks share
    the one GPU here (BENCH_REHEARSAL=1: exchange on gloo), real sweeps."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2", "--no-cpu"],
                       capture_output=True, text=True, timeout=900, env=_clean_env(BENCH_REHEARSAL="1"))
    assert r.returncode == 0, r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["latents_total"] == 8192 and out["steps"] == 5 and "rehearsal" in out
