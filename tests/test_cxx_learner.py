"""The Eigen-free C++ learners (include/moihgp_cxx/moihgp_online.hpp, moihgp_regression.hpp, lbfgsb.hpp).

CPU: the optimiser against SciPy's L-BFGS-B on bound-constrained problems; everything compiles and links.
GPU: the online objective (window / windowed mean / carried start state / proximal term, moihgp_online.h:40-93) and the
regression objective (moihgp_regression.h:34-52) against a Python restatement over the oracle; the learner end to end; and the
reference learner's call pattern written against an Eigen-like vector library (tests/cxx/eigen_like.hpp)."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, rel_err

BUILD = os.path.join(ROOT, "build")


def _cxx(name, hip_built=None):
    os.makedirs(BUILD, exist_ok=True)
    exe = os.path.join(BUILD, name)
    cmd = ["g++", "-std=c++14", "-O1", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "tests", "cxx"),
           os.path.join(ROOT, "tests", "cxx", name + ".cpp"), "-o", exe]
    if hip_built:
        libdir = os.path.dirname(hip_built)
        cmd += ["-L", libdir, "-lmoihgp", f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.run(cmd, check=True)
    return exe


def _fmt(a):
    return " ".join(repr(float(v)) for v in np.asarray(a, dtype=np.float64).ravel())


def _lines(exe, inp):
    return subprocess.run([exe], input=inp, capture_output=True, text=True, check=True).stdout.strip().split("\n")


def _arr(line):
    return np.array(line.split(), dtype=float)


# ------------------------------------------------------------------------------------------ CPU
def test_lbfgsb_matches_scipy():
    from scipy.optimize import minimize
    exe = _cxx("lbfgsb_test")

    def ros(x):
        return float(np.sum(100 * (x[1:] - x[:-1] ** 2) ** 2 + (1 - x[:-1]) ** 2))

    def rosg(x):
        g = np.zeros_like(x); a = x[1:] - x[:-1] ** 2; b = 1 - x[:-1]
        g[:-1] += -400 * a * x[:-1] - 2 * b; g[1:] += 200 * a
        return g

    def quad(x):
        n = len(x); ax = 2.5 * x.copy(); ax[1:] -= x[:-1]; ax[:-1] -= x[1:]; b = 3 * np.sin(np.arange(1, n + 1))
        return float(0.5 * x @ ax - b @ x), ax - b

    def run(prob, lb, ub, x0):
        out = _lines(exe, f"{prob} {len(x0)}\n{_fmt(lb)}\n{_fmt(ub)}\n{_fmt(x0)}\n")
        return int(out[0]), float(out[1]), _arr(out[2]), float(out[3])

    n = 10
    cases = [(-2 * np.ones(n), 2 * np.ones(n), -0.5 * np.ones(n)),                 # interior minimum
             (-2 * np.ones(n), np.r_[0.5, 2 * np.ones(n - 1)], np.zeros(n)),         # first coordinate ends on its upper bound
             (np.r_[1.2, -2 * np.ones(n - 1)], 2 * np.ones(n), np.r_[1.5, np.zeros(n - 1)])]
    for lb, ub, x0 in cases:
        it, fx, x, vhv = run(0, lb, ub, x0)
        r = minimize(ros, x0, jac=rosg, method="L-BFGS-B", bounds=list(zip(lb, ub)), options=dict(ftol=1e-15, gtol=1e-10, maxiter=5000))
        assert np.all(x >= lb) and np.all(x <= ub) and vhv > 0
        assert abs(fx - r.fun) < 1e-9 * max(1.0, abs(r.fun)) and np.abs(x - r.x).max() < 1e-5
    n = 40
    lb, ub, x0 = -0.7 * np.ones(n), 0.9 * np.ones(n), np.zeros(n)
    it, fx, x, vhv = run(1, lb, ub, x0)
    r = minimize(lambda z: quad(z)[0], x0, jac=lambda z: quad(z)[1], method="L-BFGS-B", bounds=list(zip(lb, ub)), options=dict(ftol=1e-15, gtol=1e-10))
    assert abs(fx - r.fun) < 1e-9 * abs(r.fun) and np.abs(x - r.x).max() < 1e-5 and ((x <= lb) | (x >= ub)).sum() >= 10


def test_learners_compile_and_link(hip_built):
    assert os.path.exists(_cxx("learner_test", hip_built))


def test_compat_headers_resolve_reference_includes(tmp_path):
    """`#include <moihgp/moihgp.h>` + `<moihgp/matern32ss.h>` (what the reference's C++ clients write) resolve to this library's
    class through include/moihgp_cxx/compat when that directory comes first on the include path."""
    src = tmp_path / "t.cpp"
    src.write_text('#include <moihgp/moihgp.h>\n#include <moihgp/matern32ss.h>\n#include <moihgp/matern52ss.h>\n'
                   'typedef moihgp::MOIHGP<moihgp::Matern32StateSpace> GP32;\nint main() { return sizeof(GP32) > 0 ? 0 : 1; }\n')
    subprocess.run(["g++", "-std=c++14", "-fsyntax-only", "-I", os.path.join(ROOT, "include", "moihgp_cxx", "compat"), str(src)], check=True)


# ------------------------------------------------------------------------------------------ GPU
def _params(M, L, rng):
    return np.concatenate([(np.eye(M, L) + 0.2 * rng.standard_normal((M, L))).ravel(), rng.uniform(0.5, 2, L), [0.05],
                           np.column_stack([rng.uniform(0.5, 2, L), rng.uniform(0.5, 2, L), rng.uniform(0.05, 0.2, L)]).ravel()])


def _online_objective_oracle(ref, gamma, W, p0, probe, Yticks):
    """moihgp_online.h:40-93 restated over the oracle object `ref` (already updated with p0): push_back every tick, then the
    objective at `probe` with an empty BFGS matrix (Bp = dparams)."""
    L, d, P, M = ref.L, ref.igp_dim, ref.num_igp_param, ref.M
    Y = []
    _x, _dx = np.zeros((L, d)), np.zeros((L, P, d))
    ma = np.zeros(M)
    for y in Yticks:                                              # push_back, :75-93
        Y.append(y)
        ma = np.mean(Y, axis=0)
        while len(Y) > W:
            Y.pop(0)
            _x, _dx = ref.step2(_x, Y[0] - ma, _dx)
    oldparams = ref.params.copy()                                 # :31 (getParams at construction, after update(p0))
    dparams = probe - oldparams
    ref.update(probe)                                             # :43
    loss = 0.5 * dparams @ dparams                                # :51-53
    grad = dparams.copy()
    x, dx = _x.copy(), _dx.copy()
    for yt in Y:                                                  # :61-70
        y = yt - ma
        xn, _, dxn = ref.step(x, y, dx)
        l, g = ref.negLogLikelihood(x, y, dx)
        loss += l; grad += g
        x, dx = xn, dxn
    return loss, grad, ma, len(Y)


@pytest.mark.gpu
@pytest.mark.parametrize("kern,M,L,W,nt", [("Matern32", 5, 2, 3, 7), ("Matern52", 7, 4, 4, 4), ("Matern52", 6, 3, 1, 5)])
def test_cxx_online_objective_vs_oracle(hip_built, kern, M, L, W, nt):
    from oracle import cref
    exe = _cxx("learner_test", hip_built)
    rng = np.random.default_rng(100 * M + L + W)
    ref = cref.GP(0.1, M, L, kern); ref.set_literal_ugrad(0)
    p0 = _params(M, L, rng)
    ref.update(p0)
    probe = ref.params + 0.01 * rng.standard_normal(ref.num_param)
    probe[M * L:] = np.abs(probe[M * L:])
    Y = 0.5 * rng.standard_normal((nt, M)) + 1.0
    out = _lines(exe, f"objective {M} {L} {0 if kern == 'Matern32' else 1} 0.1 0.9 {W} {nt}\n{_fmt(p0)}\n{_fmt(probe)}\n" + "\n".join(_fmt(y) for y in Y) + "\n")
    assert rel_err(_arr(out[0]), ref.params) < 1e-10
    loss, grad, ma, nY = _online_objective_oracle(ref, 0.9, W, p0, probe, list(Y))
    assert abs(float(out[1]) - loss) < 1e-9 * abs(loss)
    assert rel_err(_arr(out[2]), grad) < 1e-8
    assert rel_err(_arr(out[3]), ma) < 1e-14 and int(out[4]) == nY == min(W, nt)


@pytest.mark.gpu
@pytest.mark.parametrize("kern,M,L,nt", [("Matern32", 4, 2, 12), ("Matern52", 6, 3, 9)])
def test_cxx_regression_vs_oracle(hip_built, kern, M, L, nt):
    from oracle import cref
    exe = _cxx("learner_test", hip_built)
    rng = np.random.default_rng(7 * M + L)
    ref = cref.GP(0.1, M, L, kern); ref.set_literal_ugrad(0)
    p0 = _params(M, L, rng)
    Y = 0.5 * rng.standard_normal((nt, M))
    out = _lines(exe, f"regression {M} {L} {0 if kern == 'Matern32' else 1} 0.1 {nt}\n{_fmt(p0)}\n" + "\n".join(_fmt(y) for y in Y) + "\n")
    ref.update(p0)
    d, P = ref.igp_dim, ref.num_igp_param
    x, dx = np.zeros((L, d)), np.zeros((L, P, d))
    loss, grad = 0.0, np.zeros(ref.num_param)
    for y in Y:                                                   # moihgp_regression.h:42-50
        xn, _, dxn = ref.step(x, y, dx)
        l, g = ref.negLogLikelihood(x, y, dx)
        loss += l; grad += g
        x, dx = xn, dxn
    assert abs(float(out[0]) - loss) < 1e-9 * abs(loss) and rel_err(_arr(out[1]), grad) < 1e-8
    x = np.zeros((L, d))
    for t, y in enumerate(Y):                                     # predict, :127-139
        x, yh = ref.step(x, y)
        assert rel_err(_arr(out[2 + t]), yh) < 1e-9
    iters, lfit = out[2 + nt].split()
    assert int(iters) >= 1 and float(lfit) <= loss + 1e-9 * abs(loss)     # fit() does not increase the objective


@pytest.mark.gpu
def test_cxx_online_learner_end_to_end(hip_built):
    exe = _cxx("learner_test", hip_built)
    M, L, W, nt = 6, 3, 4, 10
    rng = np.random.default_rng(3)
    t = np.arange(nt)[:, None]
    Y = np.sin(0.3 * t + np.arange(M)[None, :]) + 0.05 * rng.standard_normal((nt, M))
    out = _lines(exe, f"online {M} {L} 0 0.1 0.9 {W} {nt} 1\n" + "\n".join(_fmt(y) for y in Y) + "\n")
    sizes = [int(v) for v in out[0].split()]
    assert sizes == [M * L + L + 1 + 3 * L, M, L, 3, 2, W]
    first = _arr(out[1])
    U0 = first[:M * L].reshape(M, L)
    assert np.abs(U0.T @ U0 - np.eye(L)).max() < 1e-10            # ctor state, moihgp.h:103-127
    assert np.allclose(first[M * L:M * L + L], 1.0) and abs(first[M * L + L] - 1e-2) < 1e-15
    yh = np.array([_arr(out[2 + k]) for k in range(nt)])
    assert np.all(np.isfinite(yh)) and rel_err(yh[0], Y[0]) < 1e-12      # first tick: zero state, y - ma = 0, yhat = ma
    pnew = _arr(out[2 + nt])
    lb = np.r_[-1e4 * np.ones(M * L), 1e-4 * np.ones(L + 1 + 3 * L)]
    ub = np.r_[1e4 * np.ones(M * L + L), 1e2 * np.ones(1 + 3 * L)]
    # (U comes back as the polar factor of the optimiser's iterate, so only the positive blocks are checked against the box)
    assert np.all(pnew[M * L:] >= lb[M * L:] - 1e-12) and np.all(pnew[M * L:] <= ub[M * L:] + 1e-12)
    fold, fnew = (float(v) for v in out[3 + nt].split())
    assert np.isfinite(fold) and np.isfinite(fnew)
    # the learner tracks: late predictions are closer to the data than the window mean alone would be far off
    assert np.abs(yh[-1] - Y[-1]).max() < 1.5


@pytest.mark.gpu
def test_cxx_reference_call_pattern_with_eigen_like_vectors(hip_built):
    from oracle import cref
    exe = _cxx("learner_test", hip_built)
    M, L = 5, 2
    rng = np.random.default_rng(11)
    ref = cref.GP(0.1, M, L, "Matern32"); ref.set_literal_ugrad(0)
    p0 = _params(M, L, rng)
    y, ma = rng.standard_normal(M), 0.3 * rng.standard_normal(M)
    out = _lines(exe, f"eigenlike {M} {L} 0 0.1\n{_fmt(p0)}\n{_fmt(y)}\n{_fmt(ma)}\n")
    ref.update(p0)
    d, P = ref.igp_dim, ref.num_igp_param
    x0, dx0 = np.zeros((L, d)), np.zeros((L, P, d))
    x1, yh = ref.step(x0, y - ma)
    loss, g = ref.negLogLikelihood(x1, y - ma, dx0)
    assert abs(float(out[0]) - loss) < 1e-9 * abs(loss)
    assert rel_err(_arr(out[1]), yh + ma) < 1e-9 and rel_err(_arr(out[2]), g) < 1e-8 and rel_err(_arr(out[3]), ref.params) < 1e-10


@pytest.mark.gpu
@pytest.mark.parametrize("kern,M,L,W,nt", [(0, 6, 3, 4, 8), (1, 9, 5, 3, 6)])
def test_cxx_device_vector_learner_follows_the_host_learner(hip_built, kern, M, L, W, nt):
    """include/moihgp_cxx/lbfgsb_dev.hpp: the same projected L-BFGS with theta, the gradient and the correction pairs on the device
    (moihgp_dvec_* kernels; objective through moihgp_update_dev / moihgp_window_eval_dev).  Same algorithm as the host learner, so from the
    same parameters the two must produce the same predictions and parameters up to the rounding order of the reductions."""
    exe = _cxx("learner_test", hip_built)
    rng = np.random.default_rng(17 + kern)
    p0 = _params(M, L, rng)
    t = np.arange(nt)[:, None]
    Y = np.sin(0.3 * t + np.arange(M)[None, :]) + 0.05 * rng.standard_normal((nt, M))
    out = _lines(exe, f"online_dev {M} {L} {kern} 0.1 0.9 {W} {nt}\n{_fmt(p0)}\n" + "\n".join(_fmt(y) for y in Y) + "\n")
    for k in range(nt):
        a, b = _arr(out[2 * k]), _arr(out[2 * k + 1])
        assert np.all(np.isfinite(a)) and rel_err(b, a) < 1e-7, k
    ph, pd = _arr(out[2 * nt]), _arr(out[2 * nt + 1])
    assert rel_err(pd, ph) < 1e-6
    assert np.abs(ph - p0).max() > 1e-4                                # the learner moved
