"""NumPy/SciPy restatement of the MOIHGP hot path -- TEST INFRASTRUCTURE ONLY.

This file is part of the parity oracle.  Nothing in the product path
(`multioutputihgp_amd/`) may import it; only `tests/`, `oracle/gen_golden.py`,
`__graft_entry__.smoke()` and `bench.py`'s cpu_baseline leg may.

PARITY UNPINNED: the reference (lim271/MultiOutputIHGP) ships no tests, golden
vectors or fixtures, and cannot be built here (needs Eigen3 >= 3.3, which is
neither vendored nor installed; no network).  This restatement therefore
follows the reference source text line by line and is cross-checked against the
independent C restatement `oracle/moihgp_oracle.c` (<= 1e-12) and analytic
anchors (closed-form Matern-3/2 expm, DARE residual).

All `file:line` citations are into /root/reference/moihgp/include/.

Third-party arithmetic the reference delegates to Eigen3 (>= 3.3, unpinned):
  * `MatrixBase::exp()` (unsupported/Eigen/MatrixFunctions, Pade + scaling and
    squaring, Higham 2005) -> here `scipy.linalg.expm` (Al-Mohy & Higham 2009);
    both are backward stable to O(eps); the C restatement carries its own Pade-13.
  * `BDCSVD` / `JacobiSVD` -> `numpy.linalg.svd` (LAPACK gesdd).  Only the polar
    factor `U V^T` (unique for full column rank) and singular values are used.
  * `ldlt().solve` -> `numpy.linalg.solve` on the SPD normal matrix.

Two reference lines are dimensionally invalid Eigen products (they abort when
Eigen assertions are on and are undefined behaviour in the reference's Release
build, `CMakeLists.txt:5` => -DNDEBUG):
  * moihgp/ihgp.h:158  `AK * AK.transpose() * dR`   ((d x d) * (1 x 1))
  * moihgp/ihgp.h:218  `-HdA[idx] * x`              ((d x 1) * (d x 1))
They are restated with their evident meaning, which is also what the sibling
lines compute: `AK * dR * AK^T` (ihgp.h:183) and the row-vector product
`(H dA) x` (HdA is stored transposed, ihgp.h:198).
"""
from __future__ import annotations

import numpy as np
from scipy.linalg import expm as _expm

DARE_TOL = 1e-8       # utils/dare.h:7
DARE_MAXITER = 100    # utils/dare.h:8


# ----------------------------------------------------------------------------
# L0: state-space models
# ----------------------------------------------------------------------------
class Matern32StateSpace:
    """moihgp/matern32ss.h:13-99 (dim 2, params = magnitude, lengthscale, noise)."""

    dim = 2
    num_param = 3

    def __init__(self):
        d, P = self.dim, self.num_param
        self.F = np.zeros((d, d)); self.F[0, 1] = 1.0            # :19-20
        self.Pinf = np.zeros((d, d))                              # :21
        self.H = np.array([[1.0, 0.0]])                           # :22-23
        self.R = np.zeros((1, 1))                                 # :24
        self.dF = [np.zeros((d, d)) for _ in range(P)]            # :25
        self.dPinf = [np.eye(d), np.zeros((d, d)), np.zeros((d, d))]   # :26-29
        self.dR = [np.zeros((1, 1)), np.zeros((1, 1)), np.ones((1, 1))]  # :30-33
        self.update(np.array([1.0, 1.0, 0.1]))                    # :34-36

    def update(self, params):                                     # :40-64
        magnitude, lengthscale = float(params[0]), float(params[1])
        lam = np.sqrt(3.0) / lengthscale
        lam2 = lam * lam
        len3 = 6.0 / (lengthscale * lengthscale * lengthscale)
        self.F[1, 0] = -lam2
        self.F[1, 1] = -2.0 * lam
        self.Pinf[0, 0] = magnitude
        self.Pinf[1, 1] = magnitude * lam2
        self.R[0, 0] = float(params[2])
        self.dF[1][1, 0] = len3
        self.dF[1][1, 1] = 2.0 * lam / lengthscale
        self.dPinf[0][1, 1] = lam2
        self.dPinf[1][1, 1] = -magnitude * len3
        self.params = np.array(params, dtype=np.float64).copy()


class Matern52StateSpace:
    """moihgp/matern52ss.h:13-110 (dim 3).  Literal, including `lam = sqrt(3)/l` (:42)."""

    dim = 3
    num_param = 3

    def __init__(self):
        d, P = self.dim, self.num_param
        self.F = np.zeros((d, d)); self.F[0, 1] = 1.0; self.F[1, 2] = 1.0   # :19-21
        self.Pinf = np.zeros((d, d))
        self.H = np.array([[1.0, 0.0, 0.0]])
        self.R = np.zeros((1, 1))
        self.dF = [np.zeros((d, d)) for _ in range(P)]
        self.dPinf = [np.zeros((d, d)) for _ in range(P)]
        self.dR = [np.zeros((1, 1)), np.zeros((1, 1)), np.ones((1, 1))]
        self.update(np.array([1.0, 1.0, 0.1]))

    def update(self, params):                                     # :38-75
        magnitude, lengthscale = float(params[0]), float(params[1])
        lam = np.sqrt(3.0) / lengthscale                          # :42 (sic)
        lam2 = lam * lam
        len2 = lengthscale * lengthscale
        len3 = len2 * lengthscale
        len4 = len2 * len2
        kappa = 5.0 / 3.0 * magnitude / len2
        kappa2 = -2.0 * kappa / lengthscale
        sq5 = np.sqrt(5.0)
        self.F[2, 0] = -lam2 * lam
        self.F[2, 1] = -3.0 * lam2
        self.F[2, 2] = -3.0 * lam
        self.Pinf[0, 0] = magnitude
        self.Pinf[2, 2] = 25.0 * magnitude / len4
        self.Pinf[1, 1] = kappa
        self.Pinf[2, 0] = -kappa
        self.Pinf[0, 2] = -kappa
        self.R[0, 0] = float(params[2])
        self.dF[1][2, 0] = 15.0 * sq5 / len4
        self.dF[1][2, 1] = 30.0 / len3
        self.dF[1][2, 2] = sq5 * lam2
        self.dPinf[0] = self.Pinf / magnitude
        self.dPinf[1][1, 1] = kappa2
        self.dPinf[1][2, 0] = -kappa2
        self.dPinf[1][0, 2] = -kappa2
        self.dPinf[1][2, 2] = -100.0 * magnitude / len2 / len3
        self.params = np.array(params, dtype=np.float64).copy()


class StackedStateSpace:
    """Sum of J independent Matern components observed through one output: the "stacked state" model of BASELINE.json's
    d=6 / d=12 configs.  NOT a class of the reference (its only models are matern32ss.h / matern52ss.h); it is the plain
    composition the reference's `IHGP<StateSpace>` template (ihgp.h:17-35) admits: any class exposing F, H, Pinf, R and
    their derivatives plugs into IHGP::update unchanged.  Built from the reference's own component classes:
        F = blockdiag(F_j), Pinf = blockdiag(Pinf_j), H = [H_1 .. H_J], R = noise,
        params = [magnitude_1, lengthscale_1, .., magnitude_J, lengthscale_J, noise]  (P = 2J + 1),
    derivative w.r.t. a component parameter = that component's derivative placed in its block, zero elsewhere."""

    base = None
    J = 1

    def __init__(self):
        self.parts = [self.base() for _ in range(self.J)]
        db = self.base.dim
        self.dim = d = db * self.J
        self.num_param = P = 2 * self.J + 1
        self.F = np.zeros((d, d)); self.Pinf = np.zeros((d, d)); self.H = np.zeros((1, d)); self.R = np.zeros((1, 1))
        self.dF = [np.zeros((d, d)) for _ in range(P)]
        self.dPinf = [np.zeros((d, d)) for _ in range(P)]
        self.dR = [np.zeros((1, 1)) for _ in range(P)]
        self.dR[P - 1][0, 0] = 1.0
        default = []
        for j in range(self.J):
            default += [1.0, float(j + 1)]
        self.update(np.array(default + [0.1]))

    def update(self, params):
        db, J = self.base.dim, self.J
        params = np.asarray(params, dtype=np.float64)
        self.R[0, 0] = params[2 * J]
        for j, part in enumerate(self.parts):
            part.update(np.array([params[2 * j], params[2 * j + 1], params[2 * J]]))
            sl = slice(j * db, (j + 1) * db)
            self.F[sl, sl] = part.F
            self.Pinf[sl, sl] = part.Pinf
            self.H[0, sl] = part.H[0]
            for q in range(2):
                self.dF[2 * j + q][sl, sl] = part.dF[q]
                self.dPinf[2 * j + q][sl, sl] = part.dPinf[q]
        self.params = params.copy()


def _stacked(base_cls, J):
    return type("%sx%d" % (base_cls.__name__, J), (StackedStateSpace,), {"base": base_cls, "J": J})


KERNELS = {"Matern32": Matern32StateSpace, "Matern52": Matern52StateSpace}
for _J in (2, 3, 4):
    KERNELS["Matern32x%d" % _J] = _stacked(Matern32StateSpace, _J)
    KERNELS["Matern52x%d" % _J] = _stacked(Matern52StateSpace, _J)


# ----------------------------------------------------------------------------
# utils/dare.h
# ----------------------------------------------------------------------------
def dare(Ad, Bd, Q, R):
    """utils/dare.h:10-33.  Returns (P, iterations, converged)."""
    P = Q.copy()
    AdT, BdT = Ad.T, Bd.T
    for it in range(DARE_MAXITER):
        G = R + BdT @ P @ Bd                                       # 1x1
        P_next = AdT @ P @ Ad - AdT @ P @ Bd @ np.linalg.inv(G) @ BdT @ P @ Ad + Q   # :23
        diff = abs(np.max(P_next - P))                             # :25 fabs(maxCoeff)
        P = (P_next + P_next.T) / 2.0                              # :26
        if diff < DARE_TOL:
            return P, it + 1, True
    return P, DARE_MAXITER, False


def dlyap(Ad, Q):
    """utils/dare.h:36-58 (literal `AdT P Ad - P + Q`, :48)."""
    P = Q.copy()
    AdT = Ad.T
    for it in range(DARE_MAXITER):
        P_next = AdT @ P @ Ad - P + Q
        diff = abs(np.max(P_next - P))
        P = (P_next + P_next.T) / 2.0
        if diff < DARE_TOL:
            return P, it + 1, True
    return P, DARE_MAXITER, False


# ----------------------------------------------------------------------------
# L1: IHGP
# ----------------------------------------------------------------------------
class IHGP:
    """moihgp/ihgp.h:17-263."""

    def __init__(self, dt, kernel="Matern32"):
        self.dt = float(dt)
        self.ss = KERNELS[kernel]() if isinstance(kernel, str) else kernel    # or any StateSpace-like object (ihgp.h:17 template)
        self.num_param = self.ss.num_param
        self.dim = self.ss.dim
        self.update(self.ss.params)                               # :33

    def update(self, params):                                     # :117-201
        ss, dt, d = self.ss, self.dt, self.dim
        ss.update(params)
        H, HT = ss.H, ss.H.T
        A = _expm(dt * ss.F)                                      # :120
        Q = ss.Pinf - A @ ss.Pinf @ A.T                           # :121
        Q = (Q + Q.T) / 2.0                                       # :122
        PP, self.dare_iters, self.dare_converged = dare(A, HT, Q, ss.R)   # :125
        S = H @ PP @ HT + ss.R                                    # :126
        K = PP @ HT / S[0, 0]                                     # :127
        PF = PP - K @ H @ PP                                      # :128
        HA = H @ A                                                # :129
        AKHA = A - K @ HA                                         # :130
        AT = A.T
        AK = A @ K
        AAKH = A - AK @ H                                         # :133
        zeros = np.zeros((d, d))
        dA, dS, dK, dAKHA, HdA = [], [], [], [], []
        self.dlyap_iters = []
        for idx in range(self.num_param):                         # :136
            dF_zero = np.array_equal(ss.dF[idx], zeros)
            dPinf_zero = np.array_equal(ss.dPinf[idx], zeros)
            dR_zero = ss.dR[idx][0, 0] == 0.0
            if dF_zero:                                           # :141
                dAi = np.zeros((d, d))
                dQ = zeros.copy() if dPinf_zero else ss.dPinf[idx] - A @ ss.dPinf[idx] @ AT   # :144-151
                if dR_zero:
                    QLyap = dQ                                    # :154
                else:
                    QLyap = AK @ ss.dR[idx] @ AK.T + dQ           # :158 (restated, see header)
            else:
                FF = np.zeros((2 * d, 2 * d))                     # :163-166
                FF[:d, :d] = ss.F
                FF[d:, d:] = ss.F
                FF[d:, :d] = ss.dF[idx]
                dAi = _expm(dt * FF)[d:, :d]                      # :167
                dAT = dAi.T
                if dPinf_zero:
                    dQ = -dAi @ ss.Pinf @ AT - A @ ss.Pinf @ dAT  # :171
                else:
                    dQ = ss.dPinf[idx] - dAi @ ss.Pinf @ AT - A @ ss.dPinf[idx] @ AT - A @ ss.Pinf @ dAT   # :175
                QLyap = dAi @ PP @ AT + A @ PP @ dAT - dAi @ PP @ HT @ AK.T - AK @ H @ PP @ dAT   # :179
                if not dR_zero:
                    QLyap = QLyap + AK @ ss.dR[idx] @ AK.T        # :183
                QLyap = QLyap + dQ
            dPP, its, _ = dlyap(AAKH, QLyap)                      # :187
            self.dlyap_iters.append(its)
            dSi = H @ dPP @ HT + ss.dR[idx]                       # :188
            dKi = (dPP - PP * dSi[0, 0] / S[0, 0]) @ HT / S[0, 0]  # :189
            if dF_zero:
                dAKHAi = -dKi @ H @ A                             # :192
                HdAi = np.zeros((d, 1))                           # :193
            else:
                dAKHAi = dAi - dKi @ H @ A - K @ H @ dAi          # :197
                HdAi = (H @ dAi).T                                # :198
            dA.append(dAi); dS.append(dSi); dK.append(dKi); dAKHA.append(dAKHAi); HdA.append(HdAi)
        self.A, self.Q, self.K, self.S, self.PF, self.HA, self.AKHA = A, Q, K, S, PF, HA, AKHA
        self.PP = PP
        self.dA, self.dS, self.dK, self.dAKHA, self.HdA = dA, dS, dK, dAKHA, HdA

    # -- step overloads -------------------------------------------------------
    def step(self, x, y=None, dx=None):
        """ihgp.h:37-100.  y=None => predict-only overload (:96-100).
        Returns (xnew, yhat[, dxnew])."""
        x = np.asarray(x, dtype=np.float64).reshape(self.dim)
        missing = y is None or np.isnan(y)
        if missing:
            xnew = self.A @ x                                     # :41/:85/:98
        else:
            xnew = self.AKHA @ x + self.K[:, 0] * y               # :50/:90
        yhat = xnew[0]
        if dx is None:
            return xnew, yhat
        dxnew = []
        for idx in range(self.num_param):
            dxi = np.asarray(dx[idx], dtype=np.float64).reshape(self.dim)
            if missing:
                dxnew.append(self.dA[idx] @ x + self.A @ dxi)     # :45
            else:
                dxnew.append(self.dAKHA[idx] @ x + self.AKHA @ dxi + self.dK[idx][:, 0] * y)   # :54
        return xnew, yhat, np.array(dxnew)

    def nll(self, x, y, dx=None):
        """ihgp.h:204-222 (uses the PRE-step state)."""
        x = np.asarray(x, dtype=np.float64).reshape(self.dim)
        S = self.S[0, 0]
        v = y - (self.HA @ x)[0]
        loss = 0.5 * (v * v / S + np.log(S))
        if dx is None:
            return loss
        grad = np.zeros(self.num_param)
        for idx in range(self.num_param):
            dxi = np.asarray(dx[idx], dtype=np.float64).reshape(self.dim)
            dv = -(self.HdA[idx][:, 0] @ x) - (self.HA @ dxi)[0]  # :218 (restated, see header)
            grad[idx] = (v * dv - 0.5 * (v * v / S - 1) * self.dS[idx][0, 0]) / S   # :219
        return loss, grad


# ----------------------------------------------------------------------------
# L2: MOIHGP
# ----------------------------------------------------------------------------
def polar_factor(Uparam):
    """moihgp.h:438-446: `svd.matrixU() * svd.matrixV().transpose()` of an M x L matrix."""
    u, _, vt = np.linalg.svd(Uparam, full_matrices=False)
    return u @ vt


class MOIHGP:
    """moihgp/moihgp.h:76-757.  Deterministic construction: U = polar(I) = I[:, :L]
    (the reference adds N(0,1e-3) noise from std::random_device, :105-125, which is
    not reproducible; callers must `update(params)` before comparing anything)."""

    def __init__(self, dt, num_output, num_latent, kernel="Matern32", threading=False):
        self.dt = float(dt)
        self.M, self.L = int(num_output), int(num_latent)
        self.threading = bool(threading) and self.L >= 2          # :128-135 (fewer than two latents: always off)
        self.igps = [IHGP(dt, kernel) for _ in range(self.L)]
        self.dim = self.igps[0].dim
        self.P = self.igps[0].num_param
        self.num_param = self.M * self.L + self.L + 1 + self.L * self.P   # :93
        self.U = polar_factor(np.eye(self.M, self.L))
        self.S = np.ones(self.L)                                  # :126
        self.sigma = 1e-2                                         # :127

    # -- params ---------------------------------------------------------------
    def update(self, params):                                     # :431-457
        params = np.asarray(params, dtype=np.float64)
        M, L, P = self.M, self.L, self.P
        sizeU = M * L
        Uparam = params[:sizeU].reshape(M, L)     # col-major (L,M) resize then transpose == row-major (M,L)
        self.U = polar_factor(Uparam)
        self.S = params[sizeU:sizeU + L].copy()
        self.sigma = float(params[sizeU + L])
        igp_params = params[sizeU + L + 1:].reshape(L, P)         # col-major (P,L) => latent-major
        for l in range(L):
            self.igps[l].update(igp_params[l])

    def get_params(self):                                         # :721-738
        return np.concatenate([
            self.U.reshape(-1), self.S, [self.sigma],
            np.concatenate([g.ss.params for g in self.igps]),
        ])

    # -- projection -----------------------------------------------------------
    def project(self, y):
        """moihgp.h:150-182: Ty = S^-1/2 U^T y, or LS over observed rows."""
        y = np.asarray(y, dtype=np.float64)
        obs = ~np.isnan(y)
        sqrtSinv = 1.0 / np.sqrt(self.S)
        if obs.sum() != self.M:                                   # :167-178
            U0 = self.U[obs]
            y0 = y[obs]
            return sqrtSinv * np.linalg.solve(U0.T @ U0, U0.T @ y0)
        return sqrtSinv * (self.U.T @ y)                          # :181

    def unproject(self, Tyhat):
        return self.U @ (np.sqrt(self.S) * Tyhat)                 # :222-225

    # -- step overloads 1..4 --------------------------------------------------
    def step(self, x, y=None, dx=None):
        """:148-428.  x [L][d], y [M] or None, dx [L][P][d] or None.
        Returns (xnew, yhat[, dxnew]).  Overload 2 (no yhat) = overload 1 minus yhat."""
        L, d, P = self.L, self.dim, self.P
        x = np.asarray(x, dtype=np.float64).reshape(L, d)
        xnew = np.zeros((L, d))
        Tyhat = np.zeros(L)
        if y is None:                                             # overload 4, :381-428
            for l in range(L):
                xnew[l], Tyhat[l] = self.igps[l].step(x[l])
            return xnew, self.unproject(Tyhat)
        Ty = self.project(y)
        if dx is None:                                            # overload 3, :304-378
            for l in range(L):
                xnew[l], Tyhat[l] = self.igps[l].step(x[l], Ty[l])
            return xnew, self.unproject(Tyhat)
        dx = np.asarray(dx, dtype=np.float64).reshape(L, P, d)    # overload 1, :148-226
        dxnew = np.zeros((L, P, d))
        for l in range(L):
            xnew[l], Tyhat[l], dxnew[l] = self.igps[l].step(x[l], Ty[l], dx[l])
        return xnew, self.unproject(Tyhat), dxnew

    # -- NLL ------------------------------------------------------------------
    def nll(self, x, y, dx=None, literal_ugrad=True):
        """:614-688 (dx None) and :460-611 (with grad)."""
        M, L, d, P = self.M, self.L, self.dim, self.P
        x = np.asarray(x, dtype=np.float64).reshape(L, d)
        y = np.asarray(y, dtype=np.float64)
        Ty = self.project(y)
        U, S, sigma = self.U, self.S, self.sigma
        y_UUTy = np.linalg.norm((np.eye(M) - U @ U.T) @ y)        # :651 / :501
        m_n = max(float(M - L), 0.0)
        loss = 0.5 * np.log(S.sum()) + 0.5 * m_n * np.log(sigma) + 0.5 * y_UUTy / sigma   # :653 / :503
        if dx is None:
            for l in range(L):
                loss += self.igps[l].nll(x[l], Ty[l])
            return loss
        dx = np.asarray(dx, dtype=np.float64).reshape(L, P, d)
        sizeU = M * L
        grad = np.zeros(self.num_param)
        sqrtSinv = 1.0 / np.sqrt(S)
        sqrtSinv3 = 1.0 / np.sqrt(S) ** 3
        pv = np.zeros(L)
        for l in range(L):                                        # :505-512 (raw y(idx), sic)
            g = self.igps[l]
            vi = y[l] - (g.HA @ x[l])[0]
            pv[l] = vi * (1 - (g.HA @ g.K)[0, 0]) / g.S[0, 0]
        if literal_ugrad:                                         # :513-552
            su, ss, svt = np.linalg.svd(U, full_matrices=False)
            sv = svt.T
            invS = np.diag(1.0 / ss)
            Lmat = np.eye(M) + su @ (invS - np.eye(L)) @ su.T
            Rmat = np.eye(L) + sv @ (invS - np.eye(L)) @ sv.T
            for idx1 in range(sizeU):
                r, c = divmod(idx1, L)
                dA = np.zeros((M, L)); dA[r, c] = 1.0
                dU = Lmat @ dA @ Rmat
                val = -(y @ U @ dU.T @ y) / sigma
                dAdT = sqrtSinv[:, None] * dU.T
                val += pv @ (dAdT @ y)
                grad[idx1] = val
        else:  # closed form for unit singular values: dU = E_rc
            Uty = U.T @ y
            grad[:sizeU] = np.outer(y, pv * sqrtSinv - Uty / sigma).reshape(-1)
        Uty = U.T @ y
        for l in range(L):                                        # :553-562
            grad[sizeU + l] = 0.5 / S[l] + pv[l] * (-0.5 * sqrtSinv3[l] * Uty[l])
        grad[sizeU + L] = 0.5 * (m_n - y_UUTy / sigma) / sigma    # :563
        igp_grad = np.zeros((L, P))
        for l in range(L):                                        # :598-606
            li, g = self.igps[l].nll(x[l], Ty[l], dx[l])
            if self.threading:                                    # :590 adds the per-latent loss; the serial branch
                loss += li                                        # :597-607 calls IHGP::negLogLikelihood and drops its value
            igp_grad[l] = g
            dn = g[P - 1]
            grad[sizeU + l] -= dn * sigma / S[l] / S[l]
            grad[sizeU + L] += dn / S[l]
        grad[sizeU + L + 1:] = igp_grad.reshape(-1)               # :608-609
        return loss, grad


# ----------------------------------------------------------------------------
# Batched (pre-projected stream) restatement of the hot loop, per latent.
# ----------------------------------------------------------------------------
def filter_stream(igp: IHGP, Ty, x0=None, dx0=None, want_grad=False):
    """Sequential sweep over one latent's projected stream Ty[T]:
    for t: (xnew, yhat_t) = step(x, Ty_t); nll_t from PRE-step x; x <- xnew
    (order as moihgp_online.h:61-70 / moihgp_regression.h:45-49).
    NaN ticks take the missing-data branch (ihgp.h:39-47) and add no NLL term
    (build-defined for the batched entry; the reference never reaches that case
    through MOIHGP, SURVEY 8a notes).
    Returns dict(yhat[T], x[d], nll, and with want_grad: dx[P][d], grad[P])."""
    T = len(Ty)
    d, P = igp.dim, igp.num_param
    x = np.zeros(d) if x0 is None else np.array(x0, dtype=np.float64)
    dx = np.zeros((P, d)) if dx0 is None else np.array(dx0, dtype=np.float64)
    yhat = np.zeros(T)
    nll = 0.0
    grad = np.zeros(P)
    for t in range(T):
        y = Ty[t]
        if want_grad:
            xnew, yh, dxnew = igp.step(x, y, dx)
            if not np.isnan(y):
                l, g = igp.nll(x, y, dx)
                nll += l; grad += g
            dx = dxnew
        else:
            xnew, yh = igp.step(x, y)
            if not np.isnan(y):
                nll += igp.nll(x, y)
        x = xnew
        yhat[t] = yh
    out = dict(yhat=yhat, x=x, nll=nll)
    if want_grad:
        out.update(dx=dx, grad=grad)
    return out
