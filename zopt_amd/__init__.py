"""zopt_amd -- MI355X-native batched LQR / iLQR / MPC solve engine.

Drop-in for the hot path of zprihoda/zopt (``zopt.lqrUtils`` / ``zopt.ilqrUtils`` /
``zopt.mpcUtils``): same module and function names, same argument meaning and return
shapes, with an optional leading batch axis.  All arithmetic runs in hand-written HIP
kernels (``zopt_amd/csrc``) behind the C ABI declared in ``include/zopt_amd.h``; there is
no CPU fallback -- if the HIP library or a GPU is missing the calls raise.
"""
from . import _lib  # noqa: F401

__all__ = ["lqrUtils", "ilqrUtils", "mpcUtils", "pytrees", "models", "quadcopter", "simulator", "jaxUtils", "dist", "io"]
__version__ = "0.1.0"
