"""Batched, device-resident entry points (include/moihgp.h part 2) over torch tensors.

torch is plumbing here: it owns device memory and streams; all arithmetic is in libmoihgp.so.

Stream layout: SERIES-MAJOR `[L, ld]` (one contiguous row per latent), fp32 or fp64.  The per-tick loop of
the reference callers (`for y in data: gp.step(x, y)`, example.py:40-42; moihgp_online.h:61-70;
moihgp_regression.h:42-50) becomes ONE call over T ticks.

Ordering (include/moihgp.h "ordering contract"): `filter`, `grad`, `project_stream`, `unproject_stream` are asynchronous on
the torch stream they are given and read the handle's tables; `LatentBank.update`, `MOIHGP.update` and `set_mixing` first wait
(on the device) for all such work already enqueued through the handle, then rewrite the tables and return when they are
complete -- an update issued behind pipelined sweeps neither overtakes them nor needs a host synchronisation from the caller.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np
import torch

from ._lib import MoihgpError, c_double_p, last_error, load_library

KERNEL_ID = {"Matern32": 0, "Matern52": 1, "Matern52ss": 1}
for _J in (2, 3, 4):          # stacked kernels (include/moihgp.h MOIHGP_STACK): "<base>x<J>", filter mode only
    KERNEL_ID["Matern32x%d" % _J] = 0 | (_J << 4)
    KERNEL_ID["Matern52x%d" % _J] = 1 | (_J << 4)
_DT = {torch.float64: 0, torch.float32: 1}


def _check(rc, lib):
    if rc != 0:
        raise MoihgpError(last_error(lib) or f"libmoihgp call failed (rc={rc})", rc)


def _stream_ptr(stream=None):
    s = torch.cuda.current_stream() if stream is None else stream
    return C.c_void_p(s.cuda_stream)


def padded_len(T: int, dtype) -> int:
    """Row length that satisfies the alignment contract of moihgp_filter_stream (16-byte vectors)."""
    epv = 2 if dtype == torch.float64 else 4
    return (T + epv - 1) // epv * epv


def alloc_stream(L: int, T: int, dtype=torch.float32, device="cuda") -> torch.Tensor:
    """[L, ld] tensor with ld = T rounded up; use `[:, :T]` for the payload."""
    return torch.empty((L, padded_len(T, dtype)), dtype=dtype, device=device)


def seg_ticks(dtype) -> int:
    """Ticks per tile of the segment-major layout: 4 KB of stream (include/moihgp.h moihgp_filter_stream_tiled)."""
    return 512 if dtype == torch.float64 else 1024


def alloc_stream_tiled(L: int, T: int, dtype=torch.float32, device="cuda") -> torch.Tensor:
    """Segment-major stream [ceil(T / SEG), L, SEG]: tile (s, l) holds ticks [s SEG, (s + 1) SEG) of latent l; the last tile is whole."""
    seg = seg_ticks(dtype)
    return torch.empty(((T + seg - 1) // seg, L, seg), dtype=dtype, device=device)


def tile_stream(Ty: torch.Tensor, T: Optional[int] = None, out: Optional[torch.Tensor] = None, stream=None) -> torch.Tensor:
    """Series-major [L, ld] -> segment-major [ceil(T / SEG), L, SEG] (ticks past T become zeros), by the library's copy kernel."""
    lib = load_library()
    L, ld = Ty.shape
    T = ld if T is None else int(T)
    if out is None:
        out = alloc_stream_tiled(L, T, Ty.dtype, Ty.device)
    _check(lib.moihgp_stream_retile(_DT[Ty.dtype], C.c_void_p(Ty.data_ptr()), C.c_void_p(out.data_ptr()), L, T, Ty.stride(0), 1, _stream_ptr(stream)), lib)
    return out


def untile_stream(Tt: torch.Tensor, T: int, out: Optional[torch.Tensor] = None, stream=None) -> torch.Tensor:
    """Segment-major [nseg, L, SEG] -> series-major [L, ld] (the first T ticks of every latent)."""
    lib = load_library()
    L = Tt.shape[1]
    if out is None:
        out = alloc_stream(L, T, Tt.dtype, Tt.device)
    _check(lib.moihgp_stream_retile(_DT[Tt.dtype], C.c_void_p(Tt.data_ptr()), C.c_void_p(out.data_ptr()), L, T, out.stride(0), 0, _stream_ptr(stream)), lib)
    return out


class LatentBank:
    """A shard of independent latent IHGPs (reference include/moihgp/ihgp.h `IHGP<SS>` x L) on one GPU.

    Holds the stationary matrices of `L` latents (computed on device by `IHGP::update`, ihgp.h:117-201)
    and runs the per-latent recursion + NLL over whole streams.
    """

    def __init__(self, dt: float, params_LP, kernel: str = "Matern52ss"):
        self._lib = load_library()
        kid = KERNEL_ID[kernel]
        J = kid >> 4
        self.stacked = J > 0
        self.P = 2 * J + 1 if J else 3
        self.d = (2 if (kid & 15) == 0 else 3) * max(J, 1)
        p = np.ascontiguousarray(np.asarray(params_LP, dtype=np.float64).reshape(-1, self.P))
        self.L = p.shape[0]
        self.kernel = kernel
        self._h = self._lib.moihgp_new_latents(KERNEL_ID[kernel], float(dt), self.L, p.ctypes.data_as(c_double_p))
        if not self._h:
            raise MoihgpError(last_error(self._lib) or "moihgp_new_latents failed")
        self.device = torch.device("cuda", torch.cuda.current_device())

    @classmethod
    def from_handle(cls, gp):
        """View the latents of a full `MOIHGP` object (pywrapper.MOIHGP) without copying."""
        self = cls.__new__(cls)
        self._lib = load_library()
        self._h = gp.handle
        self._owner = gp
        self.L = gp.num_latent
        self.d = gp.igp_dim
        self.P = gp.num_igp_param
        self.kernel = None
        self.stacked = gp.igp_dim > 3
        self.device = torch.device("cuda", torch.cuda.current_device())
        return self

    def __del__(self):
        try:
            if getattr(self, "_owner", None) is None and self._h:
                self._lib.moihgp_del(self._h)
                self._h = None
        except Exception:
            pass

    def set_option(self, name: str, value: int):
        """Per-handle tuning / test hooks (include/moihgp.h moihgp_set_option): "filter_split" (0 automatic, 1 off, n slices),
        "filter_maxlinks", "filter_variant" (tuning builds only)."""
        _check(self._lib.moihgp_set_option(self._h, name.encode(), int(value)), self._lib)

    def update(self, params_LP):
        p = np.ascontiguousarray(np.asarray(params_LP, dtype=np.float64).reshape(self.L, self.P))
        _check(self._lib.moihgp_update_latents(self._h, p.ctypes.data_as(c_double_p)), self._lib)

    def latent(self, l: int) -> dict:
        d, P = self.d, self.P
        out = dict(A=np.zeros((d, d)), K=np.zeros(d), S=np.zeros(1), HA=np.zeros(d), AKHA=np.zeros((d, d)),
                   dA=np.zeros((P, d, d)), dS=np.zeros(P), dK=np.zeros((P, d)), dAKHA=np.zeros((P, d, d)), HdA=np.zeros((P, d)))
        iters = (C.c_int * (1 + P))()
        ptrs = [out[k].ctypes.data_as(c_double_p) if k in out else None for k in ("A", "K", "S", "HA", "AKHA", "dA", "dS", "dK", "dAKHA", "HdA")]
        _check(self._lib.moihgp_get_latent(self._h, l, *ptrs, iters), self._lib)
        out["S"] = float(out["S"][0])
        out["iters"] = list(iters)
        return out

    def profile_enable(self, max_launches: int, stride: int = 1):
        """Attach HIP event pairs to filter dispatches (kernel-exact timing): to every `stride`-th one, `max_launches` pairs
        at most.  A pair costs a few microseconds of launch overlap, so a timed loop samples rather than brackets every pass."""
        _check(self._lib.moihgp_profile_enable(self._h, int(max_launches)), self._lib)
        _check(self._lib.moihgp_profile_stride(self._h, int(stride)), self._lib)
        self._prof_cap = int(max_launches)

    def profile_read(self):
        """Per-launch kernel durations (ms) of the filter dispatches since the last read."""
        n = getattr(self, "_prof_cap", 0)
        buf = (C.c_float * max(n, 1))()
        cnt = self._lib.moihgp_profile_read(self._h, buf, n)
        return [float(buf[i]) for i in range(max(cnt, 0))]

    # ---------------------------------------------------------------------------------------
    def _check_stream(self, Ty: torch.Tensor, T: Optional[int]):
        if not Ty.is_cuda or Ty.dtype not in _DT or Ty.dim() != 2 or Ty.shape[0] != self.L or Ty.stride(1) != 1:
            raise ValueError("Ty must be a CUDA tensor [L, ld] (float32/float64) with unit stride along time")
        T = Ty.shape[1] if T is None else int(T)
        if T < 0 or T > Ty.shape[1]:
            raise ValueError("T exceeds the stream tensor")
        return T, Ty.stride(0)

    @staticmethod
    def _like_stream(Ty: torch.Tensor) -> torch.Tensor:
        """Output stream with the SAME row stride as Ty (Ty may be a column slice of a wider slab)."""
        buf = torch.empty((Ty.shape[0], Ty.stride(0)), dtype=Ty.dtype, device=Ty.device)
        return buf[:, :Ty.shape[1]]

    def filter(self, Ty: torch.Tensor, T: Optional[int] = None, x: Optional[torch.Tensor] = None,
               want_yhat: bool = True, want_nll: bool = True, yhat: Optional[torch.Tensor] = None,
               nll: Optional[torch.Tensor] = None, stream=None, x_start: Optional[torch.Tensor] = None,
               nll_total: Optional[torch.Tensor] = None):
        """One sweep of ihgp.h:81-93 (+ :204-209 on the pre-step state) over T ticks for every latent.

        Returns (yhat [L, ld] or None, x [L, d] final state, nll [L] float64 or None).  Asynchronous on
        the current torch stream.  `x` (initial state) is updated IN PLACE if given.  With `x_start` the sweep starts from
        that state instead (left untouched) and `x` only receives the final state: no reset between repeated sweeps.
        `nll_total` (a 1-element float64 CUDA tensor) receives the sum of the per-latent NLLs (one-wavefront kernel queued behind the sweep)."""
        T, ld = self._check_stream(Ty, T)
        if x is None:
            x = torch.zeros((self.L, self.d), dtype=Ty.dtype, device=Ty.device)
        if want_yhat:
            if yhat is None:
                yhat = self._like_stream(Ty)
            elif (not yhat.is_cuda or yhat.dtype != Ty.dtype or yhat.dim() != 2 or yhat.stride(1) != 1
                  or yhat.shape[0] != self.L or yhat.shape[1] < T or (self.L > 1 and yhat.stride(0) < padded_len(T, Ty.dtype))
                  or yhat.stride(0) % (2 if Ty.dtype == torch.float64 else 4) != 0):
                raise ValueError("yhat must be a CUDA tensor [L, >=T] of the stream's dtype, unit stride along time, row stride a multiple of "
                                 "16 bytes and >= T rounded up to it (it need not equal the stream's: moihgp_filter_stream_v2 takes both)")
        if x.dtype != Ty.dtype or not x.is_contiguous() or tuple(x.shape) != (self.L, self.d):
            raise ValueError("x must be a contiguous [L, d] tensor of the stream dtype")
        if want_nll and nll is None:
            nll = torch.empty((self.L,), dtype=torch.float64, device=Ty.device)
        if x_start is not None and (x_start.dtype != Ty.dtype or not x_start.is_contiguous() or tuple(x_start.shape) != (self.L, self.d)):
            raise ValueError("x_start must be a contiguous [L, d] tensor of the stream dtype")
        rc = self._lib.moihgp_filter_stream_v2(
            self._h, _DT[Ty.dtype], C.c_void_p(Ty.data_ptr()), T, ld, C.c_void_p((x if x_start is None else x_start).data_ptr()),
            C.c_void_p(x.data_ptr()),
            C.c_void_p(yhat.data_ptr()) if want_yhat else None, ((yhat.stride(0) if self.L > 1 else padded_len(T, Ty.dtype)) if want_yhat else 0),
            C.c_void_p(nll.data_ptr()) if want_nll else None,
            C.c_void_p(nll_total.data_ptr()) if (nll_total is not None and want_nll) else None, _stream_ptr(stream))
        _check(rc, self._lib)
        return (yhat if want_yhat else None), x, (nll if want_nll else None)

    def filter_tiled(self, Tt: torch.Tensor, T: int, x: Optional[torch.Tensor] = None, want_yhat: bool = True, want_nll: bool = True,
                     yhat: Optional[torch.Tensor] = None, nll: Optional[torch.Tensor] = None, stream=None, x_start: Optional[torch.Tensor] = None,
                     nll_total: Optional[torch.Tensor] = None):
        """`filter` over a SEGMENT-MAJOR stream [ceil(T / SEG), L, SEG] (alloc_stream_tiled / tile_stream): same arithmetic, same results bit for
        bit; the chip reads and writes one contiguous front instead of L row streams (moihgp_filter_stream_tiled).  yhat comes back in the
        same layout.  The reference's own models only (d = 2, 3)."""
        seg = seg_ticks(Tt.dtype)
        nseg = (int(T) + seg - 1) // seg
        if not Tt.is_cuda or Tt.dtype not in _DT or not Tt.is_contiguous() or tuple(Tt.shape) != (nseg, self.L, seg):
            raise ValueError(f"Tt must be a contiguous CUDA tensor [ceil(T / {seg}) = {nseg}, L = {self.L}, {seg}]")
        if x is None:
            x = torch.zeros((self.L, self.d), dtype=Tt.dtype, device=Tt.device)
        if x.dtype != Tt.dtype or not x.is_contiguous() or tuple(x.shape) != (self.L, self.d):
            raise ValueError("x must be a contiguous [L, d] tensor of the stream dtype")
        if x_start is not None and (x_start.dtype != Tt.dtype or not x_start.is_contiguous() or tuple(x_start.shape) != (self.L, self.d)):
            raise ValueError("x_start must be a contiguous [L, d] tensor of the stream dtype")
        if want_yhat:
            if yhat is None:
                yhat = torch.empty_like(Tt)
            elif not yhat.is_cuda or yhat.dtype != Tt.dtype or not yhat.is_contiguous() or tuple(yhat.shape) != tuple(Tt.shape):
                raise ValueError("yhat must be a contiguous CUDA tensor shaped like the stream")
        if want_nll and nll is None:
            nll = torch.empty((self.L,), dtype=torch.float64, device=Tt.device)
        rc = self._lib.moihgp_filter_stream_tiled(
            self._h, _DT[Tt.dtype], C.c_void_p(Tt.data_ptr()), int(T), C.c_void_p((x if x_start is None else x_start).data_ptr()), C.c_void_p(x.data_ptr()),
            C.c_void_p(yhat.data_ptr()) if want_yhat else None, C.c_void_p(nll.data_ptr()) if want_nll else None,
            C.c_void_p(nll_total.data_ptr()) if (nll_total is not None and want_nll) else None, _stream_ptr(stream))
        _check(rc, self._lib)
        return (yhat if want_yhat else None), x, (nll if want_nll else None)

    def grad(self, Ty: torch.Tensor, T: Optional[int] = None, x: Optional[torch.Tensor] = None,
             dx: Optional[torch.Tensor] = None, want_yhat: bool = False, stream=None):
        """Sweep with sensitivities (ihgp.h:37-57) and the per-latent NLL gradient (ihgp.h:212-222).

        Returns dict(yhat, x, dx, nll [L], grad [L, P])."""
        T, ld = self._check_stream(Ty, T)
        if x is None:
            x = torch.zeros((self.L, self.d), dtype=Ty.dtype, device=Ty.device)
        if dx is None:
            dx = torch.zeros((self.L, self.P, self.d), dtype=Ty.dtype, device=Ty.device)
        if x.dtype != Ty.dtype or dx.dtype != Ty.dtype or not x.is_contiguous() or not dx.is_contiguous():
            raise ValueError("x / dx must be contiguous tensors of the stream dtype")
        yhat = self._like_stream(Ty) if want_yhat else None
        nll = torch.empty((self.L,), dtype=torch.float64, device=Ty.device)
        grad = torch.empty((self.L, self.P), dtype=torch.float64, device=Ty.device)
        rc = self._lib.moihgp_grad_stream(
            self._h, _DT[Ty.dtype], C.c_void_p(Ty.data_ptr()), T, ld, C.c_void_p(x.data_ptr()), C.c_void_p(dx.data_ptr()),
            C.c_void_p(yhat.data_ptr()) if want_yhat else None, C.c_void_p(nll.data_ptr()), C.c_void_p(grad.data_ptr()),
            _stream_ptr(stream))
        _check(rc, self._lib)
        return dict(yhat=yhat, x=x, dx=dx, nll=nll, grad=grad)


def project_stream(gp, Y: torch.Tensor, stream=None) -> torch.Tensor:
    """OILMM projection of a tick-major observation stream Y [T, M] with the mixing of `gp`
    (a pywrapper.MOIHGP): returns the series-major projected stream [L, ld] (moihgp.h:181 per tick)."""
    lib = load_library()
    if not Y.is_cuda or Y.dtype not in _DT or Y.dim() != 2 or not Y.is_contiguous() or Y.shape[1] != gp.num_output:
        raise ValueError("Y must be a contiguous CUDA tensor [T, M]")
    T = Y.shape[0]
    Ty = alloc_stream(gp.num_latent, T, Y.dtype, Y.device)
    _check(lib.moihgp_project_stream(gp.handle, _DT[Y.dtype], C.c_void_p(Y.data_ptr()), T, C.c_void_p(Ty.data_ptr()),
                                     Ty.stride(0), _stream_ptr(stream)), lib)
    return Ty


def unproject_stream(gp, Tyhat: torch.Tensor, T: int, stream=None) -> torch.Tensor:
    """Yhat [T, M] = U S^1/2 Tyhat (moihgp.h:222-225 per tick) from a series-major stream [L, ld]."""
    lib = load_library()
    Yhat = torch.empty((T, gp.num_output), dtype=Tyhat.dtype, device=Tyhat.device)
    _check(lib.moihgp_unproject_stream(gp.handle, _DT[Tyhat.dtype], C.c_void_p(Tyhat.data_ptr()), T, Tyhat.stride(0),
                                       C.c_void_p(Yhat.data_ptr()), _stream_ptr(stream)), lib)
    return Yhat
