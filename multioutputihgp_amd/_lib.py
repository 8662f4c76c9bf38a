"""ctypes loader for lib/libmoihgp.so (the C ABI declared in include/moihgp.h)."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

c_double_p = C.POINTER(C.c_double)

# every symbol include/moihgp.h declares
REFERENCE_SYMBOLS = [
    f"{pfx}_{name}"
    for pfx in ("gp32", "gp52")
    for name in ("new", "del", "step1", "step2", "step3", "step4", "update", "lik1", "lik2",
                 "get_params", "igp_dim", "num_param", "num_igp_param")
]
ADDITIVE_SYMBOLS = [
    "moihgp_last_error", "moihgp_device_count", "moihgp_version", "moihgp_new", "moihgp_del",
    "moihgp_num_output", "moihgp_num_latent", "moihgp_set_threading", "moihgp_get_threading", "moihgp_polar_iterations", "moihgp_reseed_U", "moihgp_new_latents",
    "moihgp_update_latents", "moihgp_set_mixing", "moihgp_get_latent", "moihgp_filter_stream", "moihgp_filter_stream_io", "moihgp_filter_stream_v2", "moihgp_filter_stream_tiled", "moihgp_stream_retile", "moihgp_grad_stream",
    "moihgp_project_stream", "moihgp_unproject_stream", "moihgp_stream_sync",
    "moihgp_profile_enable", "moihgp_profile_stride", "moihgp_profile_read", "moihgp_window_set", "moihgp_window_eval", "moihgp_pin_host_buffer",
    "moihgp_update_dev", "moihgp_window_eval_dev", "moihgp_update_dev_on", "moihgp_window_eval_dev_on", "moihgp_get_params_dev", "moihgp_set_option", "moihgp_release_stream",
    "moihgp_ls_shard_gram", "moihgp_ls_shard_apply",
    "moihgp_dvec_ctx_new", "moihgp_dvec_ctx_del", "moihgp_dvec_ctx_stream", "moihgp_dvec_alloc", "moihgp_dvec_alloc_mask", "moihgp_dvec_free", "moihgp_dvec_trim", "moihgp_dvec_cache_limit", "moihgp_dvec_upload", "moihgp_dvec_download",
    "moihgp_dvec_copy", "moihgp_dvec_sync", "moihgp_dvec_dot", "moihgp_dvec_axpy", "moihgp_dvec_scale", "moihgp_dvec_sub", "moihgp_dvec_clamp", "moihgp_dvec_active_set",
    "moihgp_dvec_proj_step", "moihgp_dvec_proj_grad_norm",
]


class MoihgpError(RuntimeError):
    """A failed libmoihgp call; `.rc` holds the entry's return code when it has one (include/moihgp.h: 1 invalid argument, 2 HIP
    failure, 3 unsupported input, 4 host memory)."""

    def __init__(self, msg, rc=None):
        super().__init__(msg)
        self.rc = rc


def library_path() -> str:
    # same relative location the reference loads from (pywrapper.py:22); MOIHGP_LIB points at another build of the same
    # library (kernel tuning experiments), never at a different implementation
    return os.environ.get("MOIHGP_LIB") or os.path.join(_HERE, "lib", "libmoihgp.so")


def load_library():
    """Load libmoihgp.so and declare prototypes.  Raises if the HIP library is missing:
    the product has no other implementation to fall back to."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = library_path()
    # torch bundles its own libamdhip64 (same SONAME).  Importing it first makes the dynamic loader
    # resolve our NEEDED libamdhip64.so.7 to that already-loaded copy, so tensors, streams and our
    # kernels share ONE HIP runtime.  Without torch the system ROCm runtime is used.
    try:
        import torch  # noqa: F401
    except Exception:
        pass
    if not os.path.exists(path):
        raise MoihgpError(
            f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C multioutputihgp_amd/csrc` (needs hipcc); there is no CPU fallback")
    lib = C.CDLL(path)
    for pfx in ("gp32", "gp52"):
        f = getattr(lib, f"{pfx}_new"); f.restype = C.c_void_p; f.argtypes = [C.c_double, C.c_size_t, C.c_size_t, C.c_bool]
        f = getattr(lib, f"{pfx}_del"); f.restype = None; f.argtypes = [C.c_void_p]
        for name, n in (("step1", 6), ("step2", 5), ("step3", 4), ("step4", 3), ("update", 1), ("get_params", 1)):
            f = getattr(lib, f"{pfx}_{name}"); f.restype = None; f.argtypes = [C.c_void_p] + [c_double_p] * n
        f = getattr(lib, f"{pfx}_lik1"); f.restype = C.c_double; f.argtypes = [C.c_void_p] + [c_double_p] * 4
        f = getattr(lib, f"{pfx}_lik2"); f.restype = C.c_double; f.argtypes = [C.c_void_p] + [c_double_p] * 2
        for name in ("igp_dim", "num_param", "num_igp_param"):
            f = getattr(lib, f"{pfx}_{name}"); f.restype = C.c_size_t; f.argtypes = [C.c_void_p]
    lib.moihgp_last_error.restype = C.c_char_p
    lib.moihgp_device_count.restype = C.c_int
    lib.moihgp_version.restype = C.c_int
    lib.moihgp_new.restype = C.c_void_p
    lib.moihgp_new.argtypes = [C.c_int, C.c_double, C.c_size_t, C.c_size_t]
    lib.moihgp_del.restype = None
    lib.moihgp_del.argtypes = [C.c_void_p]
    lib.moihgp_num_output.restype = C.c_size_t; lib.moihgp_num_output.argtypes = [C.c_void_p]
    lib.moihgp_num_latent.restype = C.c_size_t; lib.moihgp_num_latent.argtypes = [C.c_void_p]
    lib.moihgp_set_threading.restype = None; lib.moihgp_set_threading.argtypes = [C.c_void_p, C.c_int]
    lib.moihgp_get_threading.restype = C.c_int; lib.moihgp_get_threading.argtypes = [C.c_void_p]
    lib.moihgp_polar_iterations.restype = C.c_int; lib.moihgp_polar_iterations.argtypes = [C.c_void_p]
    lib.moihgp_reseed_U.restype = None; lib.moihgp_reseed_U.argtypes = [C.c_void_p, C.c_ulonglong]
    lib.moihgp_new_latents.restype = C.c_void_p
    lib.moihgp_new_latents.argtypes = [C.c_int, C.c_double, C.c_size_t, c_double_p]
    lib.moihgp_update_latents.restype = C.c_int
    lib.moihgp_update_latents.argtypes = [C.c_void_p, c_double_p]
    lib.moihgp_set_mixing.restype = C.c_int
    lib.moihgp_set_mixing.argtypes = [C.c_void_p, c_double_p, c_double_p, C.c_double]
    lib.moihgp_get_latent.restype = C.c_int
    lib.moihgp_get_latent.argtypes = [C.c_void_p, C.c_size_t] + [c_double_p] * 10 + [C.POINTER(C.c_int)]
    lib.moihgp_filter_stream.restype = C.c_int
    lib.moihgp_filter_stream.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.moihgp_filter_stream_io.restype = C.c_int
    lib.moihgp_filter_stream_io.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.moihgp_filter_stream_v2.restype = C.c_int
    lib.moihgp_filter_stream_v2.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.moihgp_grad_stream.restype = C.c_int
    lib.moihgp_grad_stream.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.moihgp_project_stream.restype = C.c_int
    lib.moihgp_project_stream.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]
    lib.moihgp_unproject_stream.restype = C.c_int
    lib.moihgp_unproject_stream.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p]
    lib.moihgp_profile_enable.restype = C.c_int
    lib.moihgp_profile_enable.argtypes = [C.c_void_p, C.c_int]
    lib.moihgp_profile_stride.restype = C.c_int
    lib.moihgp_profile_stride.argtypes = [C.c_void_p, C.c_int]
    lib.moihgp_profile_read.restype = C.c_int
    lib.moihgp_profile_read.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.c_int]
    lib.moihgp_window_set.restype = C.c_int
    lib.moihgp_window_set.argtypes = [C.c_void_p, c_double_p, C.c_size_t]
    lib.moihgp_window_eval.restype = C.c_int
    lib.moihgp_window_eval.argtypes = [C.c_void_p] + [c_double_p] * 6
    lib.moihgp_pin_host_buffer.restype = C.c_int
    lib.moihgp_pin_host_buffer.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    lib.moihgp_update_dev.restype = C.c_int
    lib.moihgp_update_dev.argtypes = [C.c_void_p, C.c_void_p]
    lib.moihgp_window_eval_dev.restype = C.c_int
    lib.moihgp_window_eval_dev.argtypes = [C.c_void_p] + [C.c_void_p] * 6
    lib.moihgp_filter_stream_tiled.restype = C.c_int
    lib.moihgp_filter_stream_tiled.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.moihgp_stream_retile.restype = C.c_int
    lib.moihgp_stream_retile.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int, C.c_void_p]
    lib.moihgp_update_dev_on.restype = C.c_int
    lib.moihgp_update_dev_on.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    lib.moihgp_window_eval_dev_on.restype = C.c_int
    lib.moihgp_window_eval_dev_on.argtypes = [C.c_void_p] + [C.c_void_p] * 7
    lib.moihgp_get_params_dev.restype = C.c_int
    lib.moihgp_get_params_dev.argtypes = [C.c_void_p, C.c_void_p]
    lib.moihgp_set_option.restype = C.c_int
    lib.moihgp_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_long]
    lib.moihgp_release_stream.restype = C.c_int
    lib.moihgp_release_stream.argtypes = [C.c_void_p, C.c_void_p]
    lib.moihgp_ls_shard_gram.restype = C.c_int
    lib.moihgp_ls_shard_gram.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
    lib.moihgp_ls_shard_apply.restype = C.c_int
    lib.moihgp_ls_shard_apply.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    lib.moihgp_dvec_ctx_new.restype = C.c_void_p
    lib.moihgp_dvec_ctx_del.restype = None; lib.moihgp_dvec_ctx_del.argtypes = [C.c_void_p]
    lib.moihgp_dvec_sync.restype = C.c_int; lib.moihgp_dvec_sync.argtypes = [C.c_void_p]
    lib.moihgp_dvec_trim.restype = None; lib.moihgp_dvec_trim.argtypes = []
    lib.moihgp_dvec_dot.restype = C.c_int
    lib.moihgp_dvec_dot.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_double)]
    lib.moihgp_dvec_axpy.restype = C.c_int
    lib.moihgp_dvec_axpy.argtypes = [C.c_void_p, C.c_size_t, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.moihgp_dvec_scale.restype = C.c_int
    lib.moihgp_dvec_scale.argtypes = [C.c_void_p, C.c_size_t, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.moihgp_dvec_active_set.restype = C.c_int
    lib.moihgp_dvec_active_set.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.moihgp_dvec_proj_step.restype = C.c_int
    lib.moihgp_dvec_proj_step.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_double)]
    lib.moihgp_dvec_proj_grad_norm.restype = C.c_int
    lib.moihgp_dvec_proj_grad_norm.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_double)]
    lib.moihgp_stream_sync.restype = C.c_int
    lib.moihgp_stream_sync.argtypes = [C.c_void_p]
    _LIB = lib
    return lib


def last_error(lib=None) -> str:
    lib = lib or load_library()
    msg = lib.moihgp_last_error()
    return msg.decode() if msg else ""
