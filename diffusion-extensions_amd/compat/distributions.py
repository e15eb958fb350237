"""Flat-name shim: lets scripts written against the reference's flat module layout (`from distributions import ...`)
resolve to the MI355X backend.  Put diffusion-extensions_amd/compat AND diffusion-extensions_amd on PYTHONPATH."""
from so3x.distributions import *  # noqa: F401,F403
from so3x import distributions as _impl

__all__ = list(getattr(_impl, "__all__", [n for n in dir(_impl) if not n.startswith("_")]))
from so3x.se3 import IGSO3xR3  # noqa: E402,F401  (reference distributions.py:84)
__all__ = __all__ + ["IGSO3xR3"]
