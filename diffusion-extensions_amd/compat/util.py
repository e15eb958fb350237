"""Flat-name shim: lets scripts written against the reference's flat module layout (`from util import ...`)
resolve to the MI355X backend.  Put diffusion-extensions_amd/compat AND diffusion-extensions_amd on PYTHONPATH."""
from so3x.util import *  # noqa: F401,F403
from so3x import util as _impl

__all__ = list(getattr(_impl, "__all__", [n for n in dir(_impl) if not n.startswith("_")]))
from so3x.se3 import AffineT, AffineGrad, ProtData, se3_scale, se3_lerp  # noqa: E402,F401  (reference util.py:10-59, 364-385)
__all__ = __all__ + ["AffineT", "AffineGrad", "ProtData", "se3_scale", "se3_lerp"]
