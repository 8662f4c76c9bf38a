"""Generate tests/golden/*.npz from the NumPy restatement (oracle/moihgp_numpy.py), after asserting that
the independent C restatement (oracle/moihgp_oracle.c) agrees.  TEST INFRASTRUCTURE ONLY.

The reference cannot be built or run here (Eigen3 absent), so these vectors are NOT outputs
of the reference: parity is unpinned and the fixtures pin the two restatements against each other and
against regressions.  Run:  python -m oracle.gen_golden
"""
from __future__ import annotations

import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import cref, moihgp_numpy as onp  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
SEED = 20260101


def rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    den = max(np.max(np.abs(b)), 1e-300)
    return float(np.max(np.abs(a - b)) / den)


def synth_stream(L, T, rng, nan_frac=0.0):
    """SURVEY 8d synthetic stream: sin(0.05 t (1 + l mod 7)) + 0.1 N(0,1)."""
    t = np.arange(T)[None, :]
    l = np.arange(L)[:, None]
    Ty = np.sin(0.05 * t * (1 + l % 7)) + 0.1 * rng.standard_normal((L, T))
    if nan_frac > 0:
        Ty[rng.random((L, T)) < nan_frac] = np.nan
    return Ty


def synth_params(L, rng):
    return np.column_stack([rng.uniform(0.5, 2, L), rng.uniform(0.5, 2, L), rng.uniform(0.05, 0.2, L)])


def gen_stationary():
    rng = np.random.default_rng(SEED)
    for kern in ("Matern32", "Matern52"):
        cases = [np.array([1.0, 1.0, 0.1])] + list(synth_params(7, rng))
        dts = [0.1, 0.1, 0.1, 0.05, 0.1, 0.2, 0.1, 0.1]
        out = {"params": np.array(cases), "dt": np.array(dts)}
        keys = ("A", "K", "S", "HA", "AKHA", "dA", "dS", "dK", "dAKHA", "HdA", "iters")
        acc = {k: [] for k in keys}
        for p, dt in zip(cases, dts):
            g = onp.IHGP(dt, kern)
            g.update(p)
            c = cref.ihgp_update(kern, dt, p)
            vals = dict(A=g.A, K=g.K[:, 0], S=g.S[0, 0], HA=g.HA[0], AKHA=g.AKHA, dA=np.array(g.dA),
                        dS=np.array([s[0, 0] for s in g.dS]), dK=np.array([k[:, 0] for k in g.dK]),
                        dAKHA=np.array(g.dAKHA), HdA=np.array([h[:, 0] for h in g.HdA]),
                        iters=np.array([g.dare_iters] + g.dlyap_iters))
            for k in keys[:-1]:
                cv = c.S if k == "S" else c.mat(k)
                assert rel(cv, vals[k]) < 1e-11 or np.max(np.abs(vals[k])) == 0, (kern, k, rel(cv, vals[k]))
            assert [c.dare_iters] + list(c.dlyap_iters)[:3] == list(vals["iters"])
            for k in keys:
                acc[k].append(vals[k])
        out.update({k: np.array(v) for k, v in acc.items()})
        np.savez(os.path.join(OUT, f"stationary_{kern}.npz"), **out)


def gen_moihgp():
    rng = np.random.default_rng(SEED + 1)
    for kern in ("Matern32", "Matern52"):
        for (M, L) in [(2, 1), (4, 2), (6, 6), (8, 4)]:
            # `threading` (moihgp.h:81) is observable in the value of negLogLikelihood(x, y, dx): the serial branch (:597-607, the
            # default, and forced for L < 2 by :128-135) drops the per-latent losses the threaded branch (:590) adds.
            n = onp.MOIHGP(0.1, M, L, kern)
            c = cref.GP(0.1, M, L, kern)
            nt = onp.MOIHGP(0.1, M, L, kern, threading=True)
            ct = cref.GP(0.1, M, L, kern, threading=True)
            d, P = n.dim, n.P
            params = np.concatenate([
                (np.eye(M, L) + 0.3 * rng.standard_normal((M, L))).ravel(), rng.uniform(0.5, 2, L),
                [rng.uniform(0.01, 0.1)], synth_params(L, rng).ravel()])
            n.update(params); c.update(params); nt.update(params); ct.update(params)
            x = rng.standard_normal((L, d)); dx = rng.standard_normal((L, P, d)); y = rng.standard_normal(M)
            s1 = n.step(x, y, dx); s3 = n.step(x, y); s4 = n.step(x)
            l2 = n.nll(x, y); l1, g1 = n.nll(x, y, dx)
            c1 = c.step(x, y, dx); c3 = c.step(x, y); c4 = c.step(x)
            cl2 = c.negLogLikelihood(x, y); cl1, cg1 = c.negLogLikelihood(x, y, dx)
            l1t, g1t = nt.nll(x, y, dx); cl1t, cg1t = ct.negLogLikelihood(x, y, dx)
            for a, b in list(zip(c1, s1)) + list(zip(c3, s3)) + list(zip(c4, s4)) + [(cl2, l2), (cl1, l1), (cg1, g1), (c.params, n.get_params()),
                                                                                    (cl1t, l1t), (cg1t, g1t), (ct.negLogLikelihood(x, y), l2)]:
                assert rel(a, b) < 1e-11, (kern, M, L, rel(a, b))
            assert np.array_equal(g1t, g1)                      # the flag never touches the gradient
            sum_a4 = sum(n.igps[l].nll(x[l], n.project(y)[l]) for l in range(L))
            if L >= 2:
                assert abs(l1t - l2) <= 1e-12 * abs(l2) and abs(l1 - (l2 - sum_a4)) <= 1e-12 * max(abs(l2), abs(sum_a4))
            else:
                assert l1t == l1                                  # :128-135
            out = dict(dt=0.1, M=M, L=L, params_in=params, params_out=n.get_params(), x=x, dx=dx, y=y,
                       s1_xnew=s1[0], s1_yhat=s1[1], s1_dxnew=s1[2], s3_xnew=s3[0], s3_yhat=s3[1],
                       s4_xnew=s4[0], s4_yhat=s4[1], lik2=l2, lik1=l1, lik1_threaded=l1t, sum_latent_nll=sum_a4, grad=g1)
            if M > L:   # missing-output projection, moihgp.h:167-178
                ym = y.copy(); ym[rng.integers(0, M)] = np.nan
                m3 = n.step(x, ym); cm3 = c.step(x, ym)
                assert rel(cm3[0], m3[0]) < 1e-11 and rel(cm3[1], m3[1]) < 1e-11
                out.update(y_missing=ym, m3_xnew=m3[0], m3_yhat=m3[1])
            np.savez(os.path.join(OUT, f"moihgp_{kern}_M{M}_L{L}.npz"), **out)


def gen_streams():
    rng = np.random.default_rng(SEED + 2)
    for kern in ("Matern32", "Matern52"):
        L, T = 6, 2600      # > 2 fp32 segments (1024 ticks) and > 5 fp64 segments (512), ragged tail
        params = synth_params(L, rng)
        for tag, nan_frac in (("dense", 0.0), ("nan5", 0.05)):
            Ty = synth_stream(L, T, rng, nan_frac)
            x0 = 0.1 * rng.standard_normal((L, 2 if kern == "Matern32" else 3))
            yhat = np.zeros((L, T)); xT = np.zeros_like(x0); nll = np.zeros(L)
            for l in range(L):
                g = onp.IHGP(0.1, kern); g.update(params[l])
                r = onp.filter_stream(g, Ty[l], x0=x0[l])
                yhat[l], xT[l], nll[l] = r["yhat"], r["x"], r["nll"]
            igps = cref.ihgp_array(kern, 0.1, params)
            c = cref.filter_stream(igps, Ty, x0=x0)
            assert rel(c["yhat"], yhat) < 1e-11 and rel(c["x"], xT) < 1e-11 and rel(c["nll_per_latent"], nll) < 1e-11
            np.savez(os.path.join(OUT, f"stream_{kern}_{tag}.npz"), dt=0.1, params=params, Ty=Ty, x0=x0, yhat=yhat, xT=xT, nll=nll)
        # sensitivities + gradient sweep (shorter)
        Tg = 300
        Ty = synth_stream(L, Tg, rng, 0.02)
        d = 2 if kern == "Matern32" else 3
        x0 = 0.1 * rng.standard_normal((L, d)); dx0 = 0.05 * rng.standard_normal((L, 3, d))
        yhat = np.zeros((L, Tg)); xT = np.zeros_like(x0); dxT = np.zeros_like(dx0); nll = np.zeros(L); grad = np.zeros((L, 3))
        for l in range(L):
            g = onp.IHGP(0.1, kern); g.update(params[l])
            r = onp.filter_stream(g, Ty[l], x0=x0[l], dx0=dx0[l], want_grad=True)
            yhat[l], xT[l], dxT[l], nll[l], grad[l] = r["yhat"], r["x"], r["dx"], r["nll"], r["grad"]
        c = cref.grad_stream(cref.ihgp_array(kern, 0.1, params), Ty, x0=x0, dx0=dx0)
        assert rel(c["yhat"], yhat) < 1e-11 and rel(c["dx"], dxT) < 1e-10 and rel(c["grad"], grad) < 1e-10, (rel(c["dx"], dxT), rel(c["grad"], grad))
        np.savez(os.path.join(OUT, f"gradstream_{kern}.npz"), dt=0.1, params=params, Ty=Ty, x0=x0, dx0=dx0, yhat=yhat, xT=xT, dxT=dxT, nll=nll, grad=grad)


STACKED = ("Matern32x2", "Matern52x2", "Matern52x3", "Matern52x4")


def synth_params_stacked(L, J, rng):
    cols = []
    for _ in range(J):
        cols += [rng.uniform(0.5, 2, L), rng.uniform(0.5, 2, L)]
    return np.column_stack(cols + [rng.uniform(0.05, 0.2, L)])


def gen_stacked():
    """Stacked (sum-of-Matern) latents: stationary matrices incl. sensitivities, and filter streams (dense / 5 % missing)."""
    rng = np.random.default_rng(SEED + 3)
    for kern in STACKED:
        J = int(kern[-1])
        L, T = 5, 2600
        params = synth_params_stacked(L, J, rng)
        g = onp.IHGP(0.1, kern)
        d = g.dim
        st = {k: [] for k in ("A", "K", "S", "HA", "AKHA", "dare_iters")}
        for l in range(L):
            g.update(params[l])
            c = cref.ihgp_update(kern, 0.1, params[l])
            for k, v in (("A", g.A), ("K", g.K[:, 0]), ("HA", g.HA[0]), ("AKHA", g.AKHA), ("dA", np.array(g.dA)), ("dAKHA", np.array(g.dAKHA))):
                assert rel(c.mat(k), v) < 1e-11, (kern, k, rel(c.mat(k), v))
            assert abs(c.S - g.S[0, 0]) < 1e-11 * g.S[0, 0] and c.dare_iters == g.dare_iters
            for k, v in (("A", g.A), ("K", g.K[:, 0]), ("S", g.S[0, 0]), ("HA", g.HA[0]), ("AKHA", g.AKHA), ("dare_iters", g.dare_iters)):
                st[k].append(v)
        out = dict(dt=0.1, params=params, **{k: np.array(v) for k, v in st.items()})
        for tag, nan_frac in (("dense", 0.0), ("nan5", 0.05)):
            Ty = synth_stream(L, T, rng, nan_frac)
            x0 = 0.1 * rng.standard_normal((L, d))
            yhat = np.zeros((L, T)); xT = np.zeros_like(x0); nll = np.zeros(L)
            for l in range(L):
                g.update(params[l])
                r = onp.filter_stream(g, Ty[l], x0=x0[l])
                yhat[l], xT[l], nll[l] = r["yhat"], r["x"], r["nll"]
            cres = cref.filter_stream(cref.ihgp_array(kern, 0.1, params), Ty, x0=x0)
            assert rel(cres["yhat"], yhat) < 1e-10 and rel(cres["x"], xT) < 1e-10 and rel(cres["nll_per_latent"], nll) < 1e-10
            out.update({f"{tag}_Ty": Ty, f"{tag}_x0": x0, f"{tag}_yhat": yhat, f"{tag}_xT": xT, f"{tag}_nll": nll})
        # sensitivities (ihgp.h:136-200) and a short gradient sweep with a few gaps
        P = 2 * J + 1
        sens = {k: [] for k in ("dA", "dAKHA", "dK", "dS", "HdA", "dlyap_iters")}
        for l in range(L):
            g.update(params[l])
            c = cref.ihgp_update(kern, 0.1, params[l])
            vals = dict(dA=np.array(g.dA), dAKHA=np.array(g.dAKHA), dK=np.array([k[:, 0] for k in g.dK]),
                        dS=np.array([v[0, 0] for v in g.dS]), HdA=np.array([h[:, 0] for h in g.HdA]))
            for k, v in vals.items():
                assert rel(c.mat(k), v) < 1e-10 or np.max(np.abs(v)) == 0, (kern, k, rel(c.mat(k), v))
                sens[k].append(v)
            assert list(c.dlyap_iters)[:P] == list(g.dlyap_iters)
            sens["dlyap_iters"].append(np.array(g.dlyap_iters))
        out.update({k: np.array(v) for k, v in sens.items()})
        Tg = 200
        Ty = synth_stream(L, Tg, rng, 0.02)
        x0 = 0.1 * rng.standard_normal((L, d)); dx0 = 0.05 * rng.standard_normal((L, P, d))
        yhat = np.zeros((L, Tg)); xT = np.zeros_like(x0); dxT = np.zeros_like(dx0); nll = np.zeros(L); grad = np.zeros((L, P))
        for l in range(L):
            g.update(params[l])
            r = onp.filter_stream(g, Ty[l], x0=x0[l], dx0=dx0[l], want_grad=True)
            yhat[l], xT[l], dxT[l], nll[l], grad[l] = r["yhat"], r["x"], r["dx"], r["nll"], r["grad"]
        cg = cref.grad_stream(cref.ihgp_array(kern, 0.1, params), Ty, x0=x0, dx0=dx0)
        assert rel(cg["yhat"], yhat) < 1e-10 and rel(cg["dx"], dxT) < 1e-9 and rel(cg["grad"], grad) < 1e-9, (rel(cg["dx"], dxT), rel(cg["grad"], grad))
        out.update(grad_Ty=Ty, grad_x0=x0, grad_dx0=dx0, grad_yhat=yhat, grad_xT=xT, grad_dxT=dxT, grad_nll=nll, grad_grad=grad)
        np.savez(os.path.join(OUT, f"stacked_{kern}.npz"), **out)


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    if "--moihgp-only" in sys.argv:
        gen_moihgp()
    else:
        if "--stacked-only" not in sys.argv:
            gen_stationary()
            gen_moihgp()
            gen_streams()
        gen_stacked()
    print("golden fixtures written to", OUT)
    for f in sorted(os.listdir(OUT)):
        print(f"  {f:40s} {os.path.getsize(os.path.join(OUT, f)):8d} B")
