"""multioutputihgp_amd -- MI355X (gfx950) implementation of the MOIHGP hot path.

Host-side mirror of the reference Python surface (`moihgp/__init__.py:1-7`):
`MOIHGP` is the ctypes class of `moihgp/pywrapper.py:10-270`, bound to this package's
`lib/libmoihgp.so` whose kernels are hand-written HIP.  `streams` adds the batched,
device-resident entry points (whole time streams per call) and `sharded` the
one-process-per-GPU latent sharding with an RCCL all-reduce of the likelihood.

There is no CPU fallback: importing works anywhere, constructing an object without a
usable GPU raises.
"""
from .pywrapper import MOIHGP
from .online_learning import MOIHGPOnlineLearning
from ._lib import load_library, library_path, MoihgpError

__all__ = ["MOIHGP", "MOIHGPOnlineLearning", "load_library", "library_path", "MoihgpError"]
