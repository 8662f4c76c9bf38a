"""`MOIHGP`: host-side mirror of the reference ctypes class (reference moihgp/pywrapper.py:10-270).

Same constructor, methods, properties, argument meaning and return shapes, so code written against
the reference class runs unchanged; behind it every call goes through the C ABI of
`lib/libmoihgp.so` (include/moihgp.h part 1) into HIP kernels.

Differences, all documented in INTEGRATION.md:
  * `kernel="Matern52"` works (the reference raises AttributeError, pywrapper.py:59 `self.lib`);
    it reaches `gp52_*`, which like the reference aliases the Matern-3/2 model unless
    MOIHGP_GP52_MATERN52=1.  `kernel="Matern52ss"` selects the real matern52ss.h model explicitly.
  * construction raises `MoihgpError` when no GPU / library is available (the reference would
    fail inside `cdll.LoadLibrary`).
`threading` is passed through as in the reference (pywrapper.py:12,32).  There is no thread fan-out behind it, but the
value `negLogLikelihood(x, y, dx)` returns depends on it exactly as in the reference (moihgp.h:590 vs :597-607): with the
default `threading=False` the per-latent loss terms are NOT part of the returned loss (the gradient is the same either way).
"""
from __future__ import annotations

from ctypes import c_bool, c_double, c_size_t

import numpy as np

from ._lib import MoihgpError, c_double_p, last_error, load_library


# stacked kernels (sum of J Matern components behind one output; not models of the reference): "<base>x<J>"
_STACKED = {"%sx%d" % (b, j): (i | (j << 4)) for i, b in enumerate(("Matern32", "Matern52")) for j in (2, 3, 4)}


class MOIHGP(object):

    def __init__(self, dt, num_output, num_latent, kernel="Matern32", threading=False):
        self.dt = dt
        self.__num_output = num_output
        self.__num_latent = num_latent
        lib = load_library()
        self.__lib = lib
        if kernel == "Matern32":
            pfx = "gp32"
            self.__obj = lib.gp32_new(c_double(dt), c_size_t(num_output), c_size_t(num_latent), c_bool(threading))
        elif kernel == "Matern52":
            pfx = "gp52"
            self.__obj = lib.gp52_new(c_double(dt), c_size_t(num_output), c_size_t(num_latent), c_bool(threading))
        elif kernel == "Matern52ss":
            pfx = "gp52"     # same entry points; object built with the true Matern-5/2 state space
            self.__obj = lib.moihgp_new(1, c_double(dt), c_size_t(num_output), c_size_t(num_latent))
            if self.__obj:
                lib.moihgp_set_threading(self.__obj, int(bool(threading)))
        elif kernel in _STACKED:
            pfx = "gp32"     # the entry points dispatch on the handle; the object carries the stacked state space (include/moihgp.h MOIHGP_STACK)
            self.__obj = lib.moihgp_new(_STACKED[kernel], c_double(dt), c_size_t(num_output), c_size_t(num_latent))
            if self.__obj:
                lib.moihgp_set_threading(self.__obj, int(bool(threading)))
        else:
            raise NotImplementedError("Unsupported kernel type.")
        if not self.__obj:
            raise MoihgpError(last_error(lib) or "libmoihgp: object construction failed")
        self.__del = getattr(lib, pfx + "_del")
        self.__step1 = getattr(lib, pfx + "_step1")
        self.__step2 = getattr(lib, pfx + "_step2")
        self.__step3 = getattr(lib, pfx + "_step3")
        self.__step4 = getattr(lib, pfx + "_step4")
        self.__update = getattr(lib, pfx + "_update")
        self.__lik1 = getattr(lib, pfx + "_lik1")
        self.__lik2 = getattr(lib, pfx + "_lik2")
        self.__get_params = getattr(lib, pfx + "_get_params")
        self.__num_param = int(getattr(lib, pfx + "_num_param")(self.__obj))
        self.__num_igp_param = int(getattr(lib, pfx + "_num_igp_param")(self.__obj))
        self.__igp_dim = int(getattr(lib, pfx + "_igp_dim")(self.__obj))
        # persistent staging buffers, as pywrapper.py:146-167
        self.__params = np.zeros((self.num_param,), dtype=np.float64)
        self.__params_p = self.__params.ctypes.data_as(c_double_p)
        self.__grad = np.zeros((self.num_param,), dtype=np.float64)
        self.__grad_p = self.__grad.ctypes.data_as(c_double_p)
        self.__x = np.zeros((self.num_latent, self.igp_dim), dtype=np.float64)
        self.__x_p = self.__x.ctypes.data_as(c_double_p)
        self.__y = np.zeros((self.num_output,), dtype=np.float64)
        self.__y_p = self.__y.ctypes.data_as(c_double_p)
        self.__dx = np.zeros((self.num_latent, self.num_igp_param, self.igp_dim), dtype=np.float64)
        self.__dx_p = self.__dx.ctypes.data_as(c_double_p)
        self.__xnew = np.zeros((self.num_latent, self.igp_dim), dtype=np.float64)
        self.__xnew_p = self.__xnew.ctypes.data_as(c_double_p)
        self.__yhat = np.zeros((self.num_output,), dtype=np.float64)
        self.__yhat_p = self.__yhat.ctypes.data_as(c_double_p)
        self.__dxnew = np.zeros((self.num_latent, self.num_igp_param, self.igp_dim), dtype=np.float64)
        self.__dxnew_p = self.__dxnew.ctypes.data_as(c_double_p)
        # large models: page-lock the two big persistent staging arrays (they live as long as this object) and keep a
        # persistent gradient buffer for the window objective
        self.__wgrad = np.zeros((self.num_param,), dtype=np.float64)
        if self.num_param * 8 >= (1 << 20):
            for buf in (self.__params, self.__grad, self.__wgrad):
                lib.moihgp_pin_host_buffer(self.__obj, buf.ctypes.data, buf.nbytes)

    def __del__(self):
        try:
            if self.__obj:
                self.__del(self.__obj)
                self.__obj = None
        except Exception:
            pass

    @property
    def handle(self):
        """Opaque `moihgp_gp*` for the additive batched entry points (streams.py)."""
        return self.__obj

    def step(self, x, y=None, dx=None):
        """pywrapper.py:175-196.  Returns (xnew, yhat) or (xnew, yhat, dxnew)."""
        self.__x[...] = np.asarray(x, dtype=np.float64).reshape(self.__x.shape)
        if y is None:
            self.__step4(self.__obj, self.__x_p, self.__xnew_p, self.__yhat_p)
            return self.__xnew.astype(np.float64), self.__yhat.astype(np.float64)
        self.__y[...] = np.asarray(y, dtype=np.float64).reshape(self.__y.shape)
        if dx is None:
            self.__step3(self.__obj, self.__x_p, self.__y_p, self.__xnew_p, self.__yhat_p)
            return self.__xnew.astype(np.float64), self.__yhat.astype(np.float64)
        self.__dx[...] = np.asarray(dx, dtype=np.float64).reshape(self.__dx.shape)
        self.__step1(self.__obj, self.__x_p, self.__y_p, self.__dx_p, self.__xnew_p, self.__yhat_p, self.__dxnew_p)
        return self.__xnew.astype(np.float64), self.__yhat.astype(np.float64), self.__dxnew.astype(np.float64)

    def step_no_yhat(self, x, y, dx):
        """C++ overload 2 (moihgp.h:229-301, `gp32_step2`), which the reference Python class never binds
        to a method although it loads the symbol (pywrapper.py:49)."""
        self.__x[...] = np.asarray(x, dtype=np.float64).reshape(self.__x.shape)
        self.__y[...] = np.asarray(y, dtype=np.float64).reshape(self.__y.shape)
        self.__dx[...] = np.asarray(dx, dtype=np.float64).reshape(self.__dx.shape)
        self.__step2(self.__obj, self.__x_p, self.__y_p, self.__dx_p, self.__xnew_p, self.__dxnew_p)
        return self.__xnew.astype(np.float64), self.__dxnew.astype(np.float64)

    def update(self, params):
        """pywrapper.py:199-201."""
        self.__params[...] = np.asarray(params, dtype=np.float64).reshape(self.__params.shape)
        self.__update(self.__obj, self.__params_p)

    def negLogLikelihood(self, x, y, dx=None):
        """pywrapper.py:204-222.  Returns loss or (loss, grad)."""
        self.__x[...] = np.asarray(x, dtype=np.float64).reshape(self.__x.shape)
        self.__y[...] = np.asarray(y, dtype=np.float64).reshape(self.__y.shape)
        if dx is None:
            return np.float64(self.__lik2(self.__obj, self.__x_p, self.__y_p))
        self.__dx[...] = np.asarray(dx, dtype=np.float64).reshape(self.__dx.shape)
        res = self.__lik1(self.__obj, self.__x_p, self.__y_p, self.__dx_p, self.__grad_p)
        return np.float64(res), self.__grad.astype(np.float64)

    def window_objective(self, Y, x, dx, set_window=True):
        """The learners' windowed objective in ONE call (include/moihgp.h `moihgp_window_*`): equivalent to
        `for y in Y: (xn, _, dxn) = step(x, y, dx); l, g = negLogLikelihood(x, y, dx); loss += l; grad += g; x, dx = xn, dxn`
        (online_learning.py:83-89).  Returns (loss, grad, xnew, dxnew); `grad` is a persistent buffer that the next call
        overwrites (copy it to keep it).  Ticks with missing outputs (NaN) are projected on the device by least squares over the
        observed rows like the per-tick path (and, as there, make the loss and the mixing part of the gradient NaN: moihgp.h:499-563);
        a window beyond that kernel's limits raises MoihgpError with rc == 3."""
        lib = self.__lib
        if set_window:
            Yc = np.ascontiguousarray(Y, dtype=np.float64).reshape(-1, self.num_output)
            rc = lib.moihgp_window_set(self.__obj, Yc.ctypes.data_as(c_double_p), Yc.shape[0])
            if rc != 0:
                raise MoihgpError(last_error(lib) or "moihgp_window_set failed", rc)
        xc = np.ascontiguousarray(x, dtype=np.float64).reshape(self.num_latent, self.igp_dim)
        dxc = np.ascontiguousarray(dx, dtype=np.float64).reshape(self.num_latent, self.num_igp_param, self.igp_dim)
        loss = np.zeros(1); xn = np.zeros_like(xc); dxn = np.zeros_like(dxc)
        rc = lib.moihgp_window_eval(self.__obj, xc.ctypes.data_as(c_double_p), dxc.ctypes.data_as(c_double_p), loss.ctypes.data_as(c_double_p),
                                    self.__wgrad.ctypes.data_as(c_double_p), xn.ctypes.data_as(c_double_p), dxn.ctypes.data_as(c_double_p))
        if rc != 0:
            raise MoihgpError(last_error(lib) or "moihgp_window_eval failed", rc)
        return float(loss[0]), self.__wgrad, xn, dxn      # persistent (page-locked) buffer: overwritten by the next call

    # ---- device-resident forms (include/moihgp.h "without PCIe in the optimiser's inner loop"): torch CUDA float64 tensors -----------
    @staticmethod
    def __dev_ptr(t, n, what):
        import torch
        if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float64 and t.is_contiguous() and t.numel() == n):
            raise ValueError(f"{what}: a contiguous CUDA float64 tensor of {n} elements is required")
        from ctypes import c_void_p
        return c_void_p(t.data_ptr())

    @staticmethod
    def __stream_ptr(stream, t):
        """hipStream_t the `_on` entries order themselves by: the given torch stream, else torch's current stream on the tensor's device
        -- whatever torch has queued there for the operands (the fill of a torch.zeros, a clone, a producer kernel) is waited for by an
        event on the device, not by the host (include/moihgp.h: moihgp_update_dev_on / moihgp_window_eval_dev_on)."""
        import torch
        from ctypes import c_void_p
        s = torch.cuda.current_stream(t.device) if stream is None else stream
        return c_void_p(s.cuda_stream)

    def update_dev(self, params_dev, stream=None):
        """`update` (pywrapper.py:199-201) from a parameter vector that lives on the device: no 8 (M L + ..)-byte host copy.  Ordered behind
        `stream` (default: torch's current stream); returns, like `update`, when the new tables are complete."""
        rc = self.__lib.moihgp_update_dev_on(self.__obj, self.__dev_ptr(params_dev, self.num_param, "params_dev"), self.__stream_ptr(stream, params_dev))
        if rc != 0:
            raise MoihgpError(last_error(self.__lib) or "moihgp_update_dev_on failed", rc)

    def params_dev(self, out):
        """`params` into a device vector."""
        import torch
        torch.cuda.current_stream(out.device).synchronize()          # (this entry works on the handle's stream and has no stream argument)
        rc = self.__lib.moihgp_get_params_dev(self.__obj, self.__dev_ptr(out, self.num_param, "out"))
        if rc != 0:
            raise MoihgpError(last_error(self.__lib) or "moihgp_get_params_dev failed", rc)
        return out

    def set_window(self, Y):
        """Install the window Y [W, M] (host) for `window_objective(..., set_window=False)` / `window_objective_dev`."""
        Yc = np.ascontiguousarray(Y, dtype=np.float64).reshape(-1, self.num_output)
        rc = self.__lib.moihgp_window_set(self.__obj, Yc.ctypes.data_as(c_double_p), Yc.shape[0])
        if rc != 0:
            raise MoihgpError(last_error(self.__lib) or "moihgp_window_set failed", rc)

    def window_objective_dev(self, x_dev, dx_dev, loss_dev, grad_dev, xnew_dev=None, dxnew_dev=None, stream=None):
        """`window_objective` on the installed window with every operand on the device: state in, loss (1-element tensor) and gradient
        [num_param] out, optionally the state after the window.  Asynchronous: operands are taken in `stream`'s order (default: torch's
        current stream) and the results are ordered in front of whatever is queued on it afterwards; the host is not synchronised."""
        L, d, P = self.num_latent, self.igp_dim, self.num_igp_param
        rc = self.__lib.moihgp_window_eval_dev_on(
            self.__obj, self.__dev_ptr(x_dev, L * d, "x_dev"), self.__dev_ptr(dx_dev, L * P * d, "dx_dev"), self.__dev_ptr(loss_dev, 1, "loss_dev"),
            self.__dev_ptr(grad_dev, self.num_param, "grad_dev"),
            None if xnew_dev is None else self.__dev_ptr(xnew_dev, L * d, "xnew_dev"),
            None if dxnew_dev is None else self.__dev_ptr(dxnew_dev, L * P * d, "dxnew_dev"), self.__stream_ptr(stream, x_dev))
        if rc != 0:
            raise MoihgpError(last_error(self.__lib) or "moihgp_window_eval_dev_on failed", rc)

    @property
    def num_output(self):
        return self.__num_output

    @property
    def num_latent(self):
        return self.__num_latent

    @property
    def igp_dim(self):
        return self.__igp_dim

    @property
    def num_param(self):
        return self.__num_param

    @property
    def num_igp_param(self):
        return self.__num_igp_param

    @property
    def params(self):
        self.__get_params(self.__obj, self.__params_p)
        return self.__params

    @property
    def covariance(self):
        """pywrapper.py:256-270."""
        if self.num_igp_param != 3:
            raise NotImplementedError("covariance: the reference's formula (pywrapper.py:256-270) is written for its 3-parameter Matern models")
        params = self.params.copy()
        M, L = self.num_output, self.num_latent
        U = np.reshape(params[:M * L], (M, L))
        sqrtS = np.diag(np.sqrt(params[M * L:(M + 1) * L]))
        igp_params = np.reshape(params[-L * 3:], (L, 3))
        B = np.diag([magnitude ** 0.5 * (3 ** 0.5 / lengthscale ** 0.5) ** 1.5 for magnitude, lengthscale, _ in igp_params])
        return U @ sqrtS @ B @ sqrtS @ U.T
