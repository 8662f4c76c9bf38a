"""ctypes binding of the C restatement (oracle/moihgp_oracle.c) -- TEST INFRASTRUCTURE ONLY."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
DMAX, PMAX = 3, 3
DMAX_X, PMAX_X = 12, 9            # the wide build (`make wide`), for the stacked models
MATERN32, MATERN52 = 0, 1
KERNEL_ID = {"Matern32": MATERN32, "Matern52": MATERN52}
for _J in (2, 3, 4):              # stacked models: base | (J << 4), see moihgp_oracle.h
    KERNEL_ID["Matern32x%d" % _J] = MATERN32 | (_J << 4)
    KERNEL_ID["Matern52x%d" % _J] = MATERN52 | (_J << 4)


def _fields(dmax, pmax):
    return [
        ("kernel", C.c_int), ("d", C.c_int), ("P", C.c_int),
        ("dt", C.c_double),
        ("params", C.c_double * pmax),
        ("A", C.c_double * (dmax * dmax)),
        ("Q", C.c_double * (dmax * dmax)),
        ("PP", C.c_double * (dmax * dmax)),
        ("PF", C.c_double * (dmax * dmax)),
        ("K", C.c_double * dmax),
        ("S", C.c_double),
        ("HA", C.c_double * dmax),
        ("AKHA", C.c_double * (dmax * dmax)),
        ("dA", (C.c_double * (dmax * dmax)) * pmax),
        ("dS", C.c_double * pmax),
        ("dK", (C.c_double * dmax) * pmax),
        ("dAKHA", (C.c_double * (dmax * dmax)) * pmax),
        ("HdA", (C.c_double * dmax) * pmax),
        ("dare_iters", C.c_int),
        ("dlyap_iters", C.c_int * pmax),
    ]


class _IHGPMixin:
    def mat(self, name):
        """Return field as numpy array trimmed to (d,d)/(d,)/(P,...)."""
        d, P = self.d, self.P
        a = np.ctypeslib.as_array(getattr(self, name)).copy() if name not in ("S",) else np.float64(self.S)
        if name in ("A", "Q", "PP", "PF", "AKHA"):
            return a[: d * d].reshape(d, d)
        if name in ("K", "HA"):
            return a[:d]
        if name in ("dA", "dAKHA"):
            return a[:P, : d * d].reshape(P, d, d)
        if name in ("dK", "HdA"):
            return a[:P, :d]
        if name == "dS":
            return a[:P]
        return a


class OrcIHGP(_IHGPMixin, C.Structure):
    _fields_ = _fields(DMAX, PMAX)


class OrcIHGPX(_IHGPMixin, C.Structure):          # layout of the wide build
    _fields_ = _fields(DMAX_X, PMAX_X)


_dp = C.POINTER(C.c_double)
_fp = C.POINTER(C.c_float)


def _ptr(a, typ=_dp):
    return None if a is None else a.ctypes.data_as(typ)


def _libname(native, wide):
    if os.environ.get("MOIHGP_ORACLE_ASAN") == "1":          # `make -C oracle asan-test`: the sanitizer builds stand in for both
        return "libmoihgp_oracle%s_asan.so" % ("_x" if wide else "")
    return "libmoihgp_oracle%s%s.so" % ("_x" if wide else "", "_native" if native else "")


def build(native: bool = False, wide: bool = False) -> str:
    if os.environ.get("MOIHGP_ORACLE_ASAN") == "1":
        subprocess.run(["make", "-s", "-C", _HERE, "asan"], check=True)
        return os.path.join(_HERE, "_build", _libname(native, wide))
    target = ("wide-native" if native else "wide") if wide else ("native" if native else "all")
    subprocess.run(["make", "-s", "-C", _HERE, target], check=True)
    return os.path.join(_HERE, "_build", _libname(native, wide))


_LIBS = {}


def is_wide(kernel) -> bool:
    k = KERNEL_ID[kernel] if isinstance(kernel, str) else int(kernel)
    return (k >> 4) != 0


def lib(native: bool = False, wide: bool = False):
    """native: -O3 -march=native build (CPU baseline); wide: capacity for the stacked models (OrcIHGPX structs)."""
    key = (native, wide)
    if key in _LIBS:
        return _LIBS[key]
    path = os.path.join(_HERE, "_build", _libname(native, wide))
    if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(os.path.join(_HERE, "moihgp_oracle.c")):
        build(native, wide)
    OrcIHGP = OrcIHGPX if wide else globals()["OrcIHGP"]
    L = C.CDLL(path)
    L.orc_expm.argtypes = [C.c_int, _dp, _dp]
    L.orc_ihgp_update.argtypes = [C.POINTER(OrcIHGP), C.c_int, C.c_double, _dp]
    L.orc_ihgp_update.restype = C.c_int
    L.orc_ihgp_update_ss.argtypes = [C.POINTER(OrcIHGP), C.c_int, C.c_int, C.c_double, _dp, _dp, _dp, C.c_double, _dp, _dp, _dp]
    L.orc_ihgp_update_ss.restype = C.c_int
    L.orc_ihgp_step.argtypes = [C.POINTER(OrcIHGP), _dp, C.c_int, C.c_double, _dp, _dp, _dp, _dp]
    L.orc_ihgp_nll.argtypes = [C.POINTER(OrcIHGP), _dp, C.c_double, _dp, _dp]
    L.orc_ihgp_nll.restype = C.c_double
    L.orc_gp_new.argtypes = [C.c_int, C.c_double, C.c_size_t, C.c_size_t]
    L.orc_gp_new.restype = C.c_void_p
    L.orc_gp_del.argtypes = [C.c_void_p]
    L.orc_gp_new_t.argtypes = [C.c_int, C.c_double, C.c_size_t, C.c_size_t, C.c_int]
    L.orc_gp_new_t.restype = C.c_void_p
    L.orc_gp_set_threading.argtypes = [C.c_void_p, C.c_int]
    L.orc_gp_get_threading.argtypes = [C.c_void_p]
    L.orc_gp_get_threading.restype = C.c_int
    L.orc_gp_step1.argtypes = [C.c_void_p] + [_dp] * 6
    L.orc_gp_step2.argtypes = [C.c_void_p] + [_dp] * 5
    L.orc_gp_step3.argtypes = [C.c_void_p] + [_dp] * 4
    L.orc_gp_step4.argtypes = [C.c_void_p] + [_dp] * 3
    L.orc_gp_update.argtypes = [C.c_void_p, _dp]
    L.orc_gp_lik1.argtypes = [C.c_void_p] + [_dp] * 4
    L.orc_gp_lik1.restype = C.c_double
    L.orc_gp_lik2.argtypes = [C.c_void_p] + [_dp] * 2
    L.orc_gp_lik2.restype = C.c_double
    L.orc_gp_get_params.argtypes = [C.c_void_p, _dp]
    for f in ("orc_gp_igp_dim", "orc_gp_num_param", "orc_gp_num_igp_param"):
        getattr(L, f).argtypes = [C.c_void_p]
        getattr(L, f).restype = C.c_size_t
    L.orc_gp_latent.argtypes = [C.c_void_p, C.c_size_t]
    L.orc_gp_latent.restype = C.POINTER(OrcIHGP)
    L.orc_gp_get_U.argtypes = [C.c_void_p, _dp]
    L.orc_gp_set_literal_ugrad.argtypes = [C.c_void_p, C.c_int]
    L.orc_gp_project.argtypes = [C.c_void_p, _dp, _dp]
    L.orc_polar.argtypes = [C.c_size_t, C.c_size_t, _dp, _dp, _dp]
    L.orc_polar.restype = C.c_int
    L.orc_filter_stream.argtypes = [C.POINTER(OrcIHGP), C.c_size_t, C.c_size_t, _dp, C.c_size_t, C.c_int, _dp, _dp, _dp, C.c_int]
    L.orc_filter_stream.restype = C.c_double
    L.orc_filter_stream_f32.argtypes = [C.POINTER(OrcIHGP), C.c_size_t, C.c_size_t, _fp, C.c_size_t, C.c_int, _fp, _fp, _dp, C.c_int]
    L.orc_filter_stream_f32.restype = C.c_double
    L.orc_grad_stream.argtypes = [C.POINTER(OrcIHGP), C.c_size_t, C.c_size_t, _dp, C.c_size_t, C.c_int, _dp, _dp, _dp, _dp, _dp, C.c_int]
    L.orc_grad_stream.restype = C.c_double
    L.orc_filter_stream_fast.argtypes = [C.POINTER(OrcIHGP), C.c_size_t, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p, _dp, C.c_int, C.c_int]
    L.orc_filter_stream_fast.restype = C.c_double
    L.orc_filter_stream_refshaped.argtypes = [C.POINTER(OrcIHGP), C.c_size_t, C.c_size_t, _dp, C.c_size_t, C.c_int, _dp, _dp]
    L.orc_filter_stream_refshaped.restype = C.c_double
    L.orc_max_threads.restype = C.c_int
    L.orc_dmax.restype = C.c_int
    L.orc_pmax.restype = C.c_int
    assert (L.orc_dmax(), L.orc_pmax()) == ((DMAX_X, PMAX_X) if wide else (DMAX, PMAX))
    _LIBS[key] = L
    return L


# ---------------------------------------------------------------------------------------------
def expm(A):
    A = np.ascontiguousarray(A, dtype=np.float64)
    E = np.empty_like(A)
    lib().orc_expm(A.shape[0], _ptr(A), _ptr(E))
    return E


def ihgp_update(kernel, dt, params):
    w = is_wide(kernel)
    g = OrcIHGPX() if w else OrcIHGP()
    p = np.ascontiguousarray(params, dtype=np.float64)
    rc = lib(wide=w).orc_ihgp_update(C.byref(g), KERNEL_ID[kernel] if isinstance(kernel, str) else kernel, float(dt), _ptr(p))
    assert rc >= 0
    return g


def ihgp_update_ss(ss, dt):
    """IHGP::update (ihgp.h:117-201) on an arbitrary state-space object with numpy members F, Pinf, H, R, dF, dPinf, dR."""
    d, P = ss.F.shape[0], len(ss.dF)
    w = d > DMAX or P > PMAX
    g = OrcIHGPX() if w else OrcIHGP()
    c = lambda a: np.ascontiguousarray(a, dtype=np.float64)
    F, Pinf, H = c(ss.F), c(ss.Pinf), c(ss.H).reshape(-1)
    dF, dPinf, dR = c(np.array(ss.dF)), c(np.array(ss.dPinf)), c([r[0, 0] for r in ss.dR])
    rc = lib(wide=w).orc_ihgp_update_ss(C.byref(g), d, P, float(dt), _ptr(F), _ptr(Pinf), _ptr(H), float(ss.R[0, 0]), _ptr(dF), _ptr(dPinf), _ptr(dR))
    assert rc >= 0
    return g


def ihgp_array(kernel, dt, params_LP, native=False):
    """Array of L OrcIHGP from params [L][P]."""
    params_LP = np.ascontiguousarray(params_LP, dtype=np.float64)
    L = params_LP.shape[0]
    w = is_wide(kernel)
    arr = ((OrcIHGPX if w else OrcIHGP) * L)()
    k = KERNEL_ID[kernel] if isinstance(kernel, str) else kernel
    Lb = lib(native, w)
    for l in range(L):
        Lb.orc_ihgp_update(C.byref(arr[l]), k, float(dt), _ptr(params_LP[l]))
    return arr


def polar(A):
    A = np.ascontiguousarray(A, dtype=np.float64)
    M, L = A.shape
    U = np.empty_like(A)
    sv = np.empty(L)
    rc = lib().orc_polar(M, L, _ptr(A), _ptr(U), _ptr(sv))
    if rc != 0:
        raise ValueError("orc_polar requires M >= L")
    return U, sv


class GP:
    """Mirror of the reference ctypes class (pywrapper.py:10-270) over the C oracle."""

    def __init__(self, dt, num_output, num_latent, kernel="Matern32", threading=False):
        self._L = lib(wide=is_wide(kernel))           # stacked kernels need the wide build
        self.M, self.L = num_output, num_latent
        # `threading` (pywrapper.py:12, moihgp.h:81,128-135) changes the VALUE negLogLikelihood(x, y, dx) returns: moihgp.h:590 vs :597-607
        self._h = self._L.orc_gp_new_t(KERNEL_ID[kernel], float(dt), num_output, num_latent, int(bool(threading)))
        self.igp_dim = int(self._L.orc_gp_igp_dim(self._h))
        self.num_param = int(self._L.orc_gp_num_param(self._h))
        self.num_igp_param = int(self._L.orc_gp_num_igp_param(self._h))

    def __del__(self):
        try:
            self._L.orc_gp_del(self._h)
        except Exception:
            pass

    def _c(self, a):
        return np.ascontiguousarray(a, dtype=np.float64)

    def step(self, x, y=None, dx=None):
        x = self._c(x)
        xnew = np.zeros((self.L, self.igp_dim))
        yhat = np.zeros(self.M)
        if y is None:
            self._L.orc_gp_step4(self._h, _ptr(x), _ptr(xnew), _ptr(yhat))
            return xnew, yhat
        y = self._c(y)
        if dx is None:
            self._L.orc_gp_step3(self._h, _ptr(x), _ptr(y), _ptr(xnew), _ptr(yhat))
            return xnew, yhat
        dx = self._c(dx)
        dxnew = np.zeros((self.L, self.num_igp_param, self.igp_dim))
        self._L.orc_gp_step1(self._h, _ptr(x), _ptr(y), _ptr(dx), _ptr(xnew), _ptr(yhat), _ptr(dxnew))
        return xnew, yhat, dxnew

    def step2(self, x, y, dx):
        x, y, dx = self._c(x), self._c(y), self._c(dx)
        xnew = np.zeros((self.L, self.igp_dim))
        dxnew = np.zeros((self.L, self.num_igp_param, self.igp_dim))
        self._L.orc_gp_step2(self._h, _ptr(x), _ptr(y), _ptr(dx), _ptr(xnew), _ptr(dxnew))
        return xnew, dxnew

    def update(self, params):
        p = self._c(params)
        assert p.size == self.num_param
        self._L.orc_gp_update(self._h, _ptr(p))

    def negLogLikelihood(self, x, y, dx=None):
        x, y = self._c(x), self._c(y)
        if dx is None:
            return float(self._L.orc_gp_lik2(self._h, _ptr(x), _ptr(y)))
        dx = self._c(dx)
        grad = np.zeros(self.num_param)
        loss = float(self._L.orc_gp_lik1(self._h, _ptr(x), _ptr(y), _ptr(dx), _ptr(grad)))
        return loss, grad

    @property
    def params(self):
        p = np.zeros(self.num_param)
        self._L.orc_gp_get_params(self._h, _ptr(p))
        return p

    @property
    def U(self):
        U = np.zeros((self.M, self.L))
        self._L.orc_gp_get_U(self._h, _ptr(U))
        return U

    @property
    def threading(self):
        return bool(self._L.orc_gp_get_threading(self._h))

    def latent(self, l) -> OrcIHGP:
        return self._L.orc_gp_latent(self._h, l).contents

    def project(self, y):
        y = self._c(y)
        Ty = np.zeros(self.L)
        self._L.orc_gp_project(self._h, _ptr(y), _ptr(Ty))
        return Ty

    def set_literal_ugrad(self, flag):
        self._L.orc_gp_set_literal_ugrad(self._h, int(bool(flag)))


def filter_stream(igps, Ty, ld=None, layout=0, x0=None, want_yhat=True, nthreads=1, native=False, yhat_out=None):
    """Ty: [L][T] (layout 0) or [T][L] (layout 1) float64 or float32.  yhat_out: a buffer like Ty to write the means into (timing loops: a fresh
    164 MB array per call costs more in page faults than the sweep itself on 16 threads)."""
    Lb = lib(native, isinstance(igps[0], OrcIHGPX))
    f32 = Ty.dtype == np.float32
    Ty = np.ascontiguousarray(Ty)
    if layout == 0:
        L, T = Ty.shape
    else:
        T, L = Ty.shape
    ld = Ty.shape[1] if ld is None else ld
    d = igps[0].d
    dt = np.float32 if f32 else np.float64
    x = np.zeros((L, d), dtype=dt) if x0 is None else np.array(x0, dtype=dt).reshape(L, d).copy()
    yhat = (np.zeros_like(Ty) if yhat_out is None else yhat_out) if want_yhat else None
    assert yhat is None or (yhat.dtype == Ty.dtype and yhat.shape == Ty.shape and yhat.flags["C_CONTIGUOUS"])
    nll_l = np.zeros(L)
    if f32:
        nll = Lb.orc_filter_stream_f32(igps, L, T, _ptr(Ty, _fp), ld, layout, _ptr(x, _fp), _ptr(yhat, _fp), _ptr(nll_l), nthreads)
    else:
        nll = Lb.orc_filter_stream(igps, L, T, _ptr(Ty), ld, layout, _ptr(x), _ptr(yhat), _ptr(nll_l), nthreads)
    return dict(yhat=yhat, x=x, nll=float(nll), nll_per_latent=nll_l)


def filter_stream_fast(igps, Ty, ld=None, layout=0, x0=None, want_yhat=True, nthreads=1, native=False, yhat_out=None):
    """The fair-optimised CPU baseline (orc_filter_stream_fast: d-specialised, SIMD across latents); same contract as filter_stream."""
    Lb = lib(native, isinstance(igps[0], OrcIHGPX))
    f32 = Ty.dtype == np.float32
    Ty = np.ascontiguousarray(Ty)
    L, T = Ty.shape if layout == 0 else Ty.shape[::-1]
    ld = Ty.shape[1] if ld is None else ld
    d = igps[0].d
    dt = np.float32 if f32 else np.float64
    x = np.zeros((L, d), dtype=dt) if x0 is None else np.array(x0, dtype=dt).reshape(L, d).copy()
    yhat = (np.zeros_like(Ty) if yhat_out is None else yhat_out) if want_yhat else None
    assert yhat is None or (yhat.dtype == Ty.dtype and yhat.shape == Ty.shape and yhat.flags["C_CONTIGUOUS"])
    nll_l = np.zeros(L)
    nll = Lb.orc_filter_stream_fast(igps, L, T, Ty.ctypes.data, ld, layout, x.ctypes.data, yhat.ctypes.data if want_yhat else None, _ptr(nll_l), nthreads, int(f32))
    return dict(yhat=yhat, x=x, nll=float(nll), nll_per_latent=nll_l)


def grad_stream(igps, Ty, layout=0, x0=None, dx0=None, want_yhat=True, nthreads=1):
    Lb = lib(wide=isinstance(igps[0], OrcIHGPX))
    Ty = np.ascontiguousarray(Ty, dtype=np.float64)
    if layout == 0:
        L, T = Ty.shape
    else:
        T, L = Ty.shape
    d, P = igps[0].d, igps[0].P
    x = np.zeros((L, d)) if x0 is None else np.array(x0, dtype=np.float64).reshape(L, d).copy()
    dx = np.zeros((L, P, d)) if dx0 is None else np.array(dx0, dtype=np.float64).reshape(L, P, d).copy()
    yhat = np.zeros_like(Ty) if want_yhat else None
    nll_l = np.zeros(L)
    grad = np.zeros((L, P))
    nll = Lb.orc_grad_stream(igps, L, T, _ptr(Ty), Ty.shape[1], layout, _ptr(x), _ptr(dx), _ptr(yhat), _ptr(nll_l), _ptr(grad), nthreads)
    return dict(yhat=yhat, x=x, dx=dx, nll=float(nll), nll_per_latent=nll_l, grad=grad)
