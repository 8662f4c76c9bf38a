"""The online learner mirror (multioutputihgp_amd/online_learning.py) against golden vectors produced by the reference's OWN
unmodified Python files running on the oracle shim (oracle/gen_golden_learner.py).  CPU: the mirror over the oracle backend
(pins the host logic).  GPU: the mirror over the HIP library (pins the whole path)."""
import numpy as np
import pytest

from conftest import load_golden, rel_err


class OracleBackend:
    """MOIHGP-compatible object over the C oracle (test infrastructure)."""

    def __init__(self, dt, num_output, num_latent, kernel="Matern32", threading=False):
        from oracle import cref
        self._gp = cref.GP(dt, num_output, num_latent, kernel, threading=threading)
        self.num_output, self.num_latent = num_output, num_latent
        self.igp_dim, self.num_param, self.num_igp_param = self._gp.igp_dim, self._gp.num_param, self._gp.num_igp_param

    def step(self, x, y=None, dx=None):
        return self._gp.step(x, y, dx)

    def negLogLikelihood(self, x, y, dx=None):
        return self._gp.negLogLikelihood(x, y, dx)

    def update(self, p):
        self._gp.update(p)

    @property
    def params(self):
        return self._gp.params


def run_learner(g, backend=None):
    from multioutputihgp_amd.online_learning import MOIHGPOnlineLearning
    M, L = int(g["M"]), int(g["L"])
    # `threading` (online_learning.py:12) decides whether the objective VALUE holds the per-latent losses (moihgp.h:590 vs :597-607)
    learner = MOIHGPOnlineLearning(float(g["dt"]), M, L, float(g["gamma"]), windowsize=int(g["W"]), kernel="Matern32", threading=bool(g["threading"]),
                                   backend=backend)
    learner.moihgp.update(g["p0"])            # the reference's ctor draws a random U; start from the golden's parameters
    yhat, params = [], []
    for y in g["Y"]:
        yhat.append(learner.step(y.copy()))
        params.append(learner.params.copy())
    return np.array(yhat), np.array(params)


CASES = ["a", "b", "c", "at", "bt"]          # "at" / "bt": the streams of "a" / "b" with threading=True


def test_threading_changes_the_learner_trajectory():
    """The flag is not 'speed only': the golden trajectories of the reference's own learner differ between its two values."""
    a, at = load_golden("learner_a.npz"), load_golden("learner_at.npz")
    assert np.array_equal(a["Y"], at["Y"]) and not bool(a["threading"]) and bool(at["threading"])
    assert rel_err(at["params"][-1], a["params"][-1]) > 1e-3


@pytest.mark.parametrize("case", CASES)
def test_learner_logic_over_oracle_backend(case):
    g = load_golden(f"learner_{case}.npz")
    yhat, params = run_learner(g, backend=OracleBackend)
    assert rel_err(yhat, g["yhat"]) < 1e-9
    assert rel_err(params, g["params"]) < 1e-8


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES)
def test_learner_over_hip_library(hip_built, case):
    g = load_golden(f"learner_{case}.npz")
    yhat, params = run_learner(g)
    # L-BFGS-B amplifies rounding differences of the objective along its 5 iterations per tick
    assert rel_err(yhat, g["yhat"]) < 1e-6
    assert rel_err(params, g["params"]) < 1e-5
