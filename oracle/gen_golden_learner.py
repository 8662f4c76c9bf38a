"""Golden vectors for the online learner from the reference's OWN Python files, run unmodified.

TEST INFRASTRUCTURE ONLY.  The reference's `moihgp/online_learning.py` + `pywrapper.py` are pure Python over a ctypes
library `lib/libmoihgp.so` that cannot be built here (Eigen3 absent).  This script makes a package directory of SYMLINKS to
the reference's three Python files (nothing of the reference is copied, `pywrapper.py:22` locates its library next to
`__file__`, which for a symlink is the link's own directory), puts a shim library next to them that exports the reference's
gp32_* symbols on top of the C oracle (oracle/refshim.c + moihgp_oracle.c), and drives `MOIHGPOnlineLearning.step` on a seeded
stream.  What is pinned is therefore the reference's learner LOGIC (EMA de-meaning, window advance, proximal term, SciPy
L-BFGS-B call, online_learning.py:53-105) with the oracle's arithmetic underneath -- for both values of `threading`
(online_learning.py:12), which decides whether the objective value SciPy's line search sees contains the per-latent losses
(moihgp.h:590 vs :597-607; the gradient is the same).  Only data (inputs + expected outputs) is committed:
tests/golden/learner_*.npz.  Needs /root/reference (this container only).   Run:  python -m oracle.gen_golden_learner
"""
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = "/root/reference/moihgp"
OUT = os.path.join(ROOT, "tests", "golden")


def make_stream(seed, T, M):
    rng = np.random.default_rng(seed)
    t = np.arange(T)[:, None] * 0.1
    base = np.concatenate([np.sin(t * (1 + 0.3 * np.arange(M // 2))), np.cos(0.7 * t * (1 + 0.2 * np.arange(M - M // 2)))], axis=1)
    return base + 0.05 * rng.standard_normal((T, M)) + np.linspace(-1, 1, M)[None, :]


def main():
    tmp = tempfile.mkdtemp(prefix="moihgp_ref_")
    try:
        pkg = os.path.join(tmp, "moihgp")
        os.makedirs(os.path.join(pkg, "lib"))
        for f in ("__init__.py", "pywrapper.py", "online_learning.py"):
            os.symlink(os.path.join(REF, f), os.path.join(pkg, f))      # a link, not a copy
        subprocess.run(["gcc", "-O2", "-std=c99", "-fPIC", "-shared", "-ffp-contract=off", "-fopenmp", "-o", os.path.join(pkg, "lib", "libmoihgp.so"),
                        os.path.join(HERE, "refshim.c"), os.path.join(HERE, "moihgp_oracle.c"), "-I", HERE, "-lm"], check=True)
        sys.path.insert(0, tmp)
        import moihgp as ref                                             # the reference package, unmodified
        cases = {"a": (4, 2, 14, 3, 0.9, None, False), "b": (6, 3, 10, 1, 0.5, None, False), "c": (4, 2, 10, 2, 0.9, (4, 1), False),
                 "at": (4, 2, 14, 3, 0.9, None, True), "bt": (6, 3, 10, 1, 0.5, None, True)}      # same streams, threading on
        for name, (M, L, T, W, gamma, nan_at, threading) in cases.items():
            Y = make_stream(11 + M, T, M)
            if nan_at:
                Y[nan_at] = np.nan
            learner = ref.MOIHGPOnlineLearning(0.1, M, L, gamma, windowsize=W, kernel="Matern32", threading=threading)
            p0 = learner.params.copy()
            yhat, params = [], []
            for y in Y:
                yhat.append(learner.step(y.copy()).copy())
                params.append(learner.params.copy())
            np.savez(os.path.join(OUT, f"learner_{name}.npz"), dt=0.1, M=M, L=L, W=W, gamma=gamma, threading=threading, Y=Y, p0=p0, yhat=np.array(yhat), params=np.array(params))
            print(name, "yhat[-1] =", yhat[-1][:3], " |dparams| =", np.linalg.norm(params[-1] - p0))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
