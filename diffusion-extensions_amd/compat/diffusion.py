"""Flat-name shim: lets scripts written against the reference's flat module layout (`from diffusion import ...`)
resolve to the MI355X backend.  Put diffusion-extensions_amd/compat AND diffusion-extensions_amd on PYTHONPATH."""
from so3x.diffusion import *  # noqa: F401,F403
from so3x import diffusion as _impl

__all__ = list(getattr(_impl, "__all__", [n for n in dir(_impl) if not n.startswith("_")]))
from so3x.se3 import SE3Diffusion, ProjectedSE3Diffusion  # noqa: E402,F401  (reference diffusion.py:432, 525)
__all__ = __all__ + ["SE3Diffusion", "ProjectedSE3Diffusion"]
