"""Flat-name shim: lets scripts written against the reference's flat module layout (`from so3_lock_test import ...`)
resolve to the MI355X backend.  Put diffusion-extensions_amd/compat AND diffusion-extensions_amd on PYTHONPATH."""
from so3x.so3_lock_test import *  # noqa: F401,F403
from so3x import so3_lock_test as _impl

__all__ = list(getattr(_impl, "__all__", [n for n in dir(_impl) if not n.startswith("_")]))
