"""`MOIHGPOnlineLearning`: host-side mirror of the reference's Python online learner
(reference moihgp/online_learning.py:10-115) on top of the HIP-backed `MOIHGP`.

Same constructor, `step(y)`, `covariance`, `params`.  Per tick (online_learning.py:53-105):
  1. exponential moving average of the observations (weight 1/2; a missing output keeps drifting by its last
     increment) used to de-mean every tick;
  2. the window buffer grows by the tick; when it exceeds `windowsize` the oldest tick is dropped and the carried
     window-start state (x, dx) advances by one step on the NEW front tick;
  3. one filter step with sensitivities gives the prediction for this tick;
  4. hyper-parameters are re-fitted by L-BFGS-B (SciPy, <= 5 iterations, <= 3 line-search steps, box bounds) on
     gamma/2 * dtheta^T H^-1 dtheta + sum over the window of the NLL, H^-1 being the previous solve's
     inverse-Hessian estimate.
The optimiser stays on the host (BASELINE.json north_star); the window sum -- the hot loop -- is ONE device call
(`MOIHGP.window_objective`, include/moihgp.h `moihgp_window_*`), missing outputs included (least-squares projection of such
ticks on the device, moihgp.h:485-494); only a window beyond that kernel's limits (more than 64 outputs missing in one tick, fewer
observed outputs than latents) falls back to the reference's tick-by-tick loop.

The proximal term needs the dense inverse Hessian of all M*L + L + 1 + 3L parameters, as in the reference; that is a
small-model construct (it is quadratic in the parameter count).
"""
from __future__ import annotations

import numpy as np
from scipy.optimize import minimize

from .pywrapper import MOIHGP


class MOIHGPOnlineLearning:

    def __init__(self, dt, num_output, num_latent, gamma, x_init=None, windowsize=None, kernel="Matern32", threading=False,
                 backend=None):
        # `backend`: optional factory with the MOIHGP constructor signature (dependency injection for tests)
        make = MOIHGP if backend is None else backend
        self.moihgp = make(dt, num_output, num_latent, kernel=kernel, threading=threading)
        self.num_output, self.num_latent = num_output, num_latent
        self.ihgp_dim = self.moihgp.igp_dim
        self.ihgp_nparam = self.moihgp.num_igp_param
        inf = np.inf
        self.parameter_bounds = ([(-inf, inf)] * (num_output * num_latent)          # mixing matrix
                                 + [(1e-4, inf)] * num_latent                        # latent scales S
                                 + [(1e-4, 1e2)]                                     # mixing noise sigma
                                 + [(1e-2, 1e2), (1e-2, 1e2), (1e-4, 1e2)] * num_latent)   # magnitude, lengthscale, noise
        self.gamma = gamma
        zeros_x = np.zeros((num_latent, self.ihgp_dim))
        self.x = zeros_x.copy() if x_init is None else x_init
        self.xinit = zeros_x.copy() if x_init is None else x_init
        self.dx = np.zeros((num_latent, self.ihgp_nparam, self.ihgp_dim))
        self.dxinit = np.zeros_like(self.dx)
        self.hess_inv = np.eye(len(self.moihgp.params))
        self.buffer = []
        self.windowsize = 1 if windowsize is None else windowsize
        self.ma = None
        self.dma = np.zeros(num_output)

    # -- pieces of one tick ---------------------------------------------------------------------------
    def _update_mean(self, y):
        if self.ma is None:
            self.ma = np.where(np.isnan(y), 0.0, y).astype(np.float64)
            return
        previous = self.ma.copy()
        missing = np.isnan(y)
        self.ma = np.where(missing, previous + self.dma, 0.5 * y + 0.5 * previous)
        self.dma = self.ma - previous

    def _advance_window(self, y):
        self.buffer.append(y)
        while len(self.buffer) > self.windowsize:
            self.buffer.pop(0)
            self.xinit, _, self.dxinit = self.moihgp.step(self.xinit, y=self.buffer[0] - self.ma, dx=self.dxinit)

    def _window_sum(self, want_grad):
        """sum over the buffered window of the NLL (and its gradient) from the carried window-start state."""
        Yw = np.array(self.buffer) - self.ma
        gp = self.moihgp
        if want_grad and hasattr(gp, "window_objective"):
            try:
                loss, grad, _, _ = gp.window_objective(Yw, self.xinit, self.dxinit)
                return loss, grad
            except RuntimeError as e:
                if getattr(e, "rc", None) != 3:          # 3: missing outputs beyond the batched kernel's limits -> the loop below
                    raise
        xt, dxt = self.xinit, self.dxinit
        loss, grad = 0.0, (np.zeros(gp.num_param) if want_grad else None)
        for yt in Yw:
            xnext, _, dxnext = gp.step(xt, y=yt, dx=dxt)
            if want_grad:
                l, g = gp.negLogLikelihood(xt, yt, dxt)
                grad += g
            else:
                l = gp.negLogLikelihood(xt, yt)
            loss += l
            xt, dxt = xnext, dxnext
        return loss, grad

    # -- public ---------------------------------------------------------------------------------------
    def step(self, y=None):
        y = np.asarray(y, dtype=np.float64)
        self._update_mean(y)
        self._advance_window(y)
        xnew, yhat, dxnew = self.moihgp.step(self.x, y=y - self.ma, dx=self.dx)
        yhat = yhat + self.ma
        self.x, self.dx = xnew, dxnew

        theta0 = self.moihgp.params.copy()

        def objective(theta):
            dtheta = theta - theta0
            self.moihgp.update(theta)
            p = np.linalg.solve(self.hess_inv, dtheta)
            wl, wg = self._window_sum(True)
            return self.gamma * 0.5 * dtheta.dot(p) + wl, self.gamma * p + wg

        res = minimize(objective, theta0, jac=True, method="L-BFGS-B", bounds=self.parameter_bounds,
                       options=dict(maxiter=5, maxls=3))
        self.moihgp.update(res.x)
        self.hess_inv = res.hess_inv.todense()
        return yhat

    @property
    def covariance(self):
        return self.moihgp.covariance

    @property
    def params(self):
        return self.moihgp.params
