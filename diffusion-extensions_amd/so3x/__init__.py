"""so3x -- MI355X-native SO(3) diffusion hot path (drop-in for the reference's
diffusion.SO3Diffusion / distributions.IsotropicGaussianSO3 / util rotation algebra /
so3_train.RotPredict).  All compute runs in libso3x.so (hand-written HIP, gfx950)."""
from . import backend  # noqa: F401
from .rng import manual_seed  # noqa: F401

__all__ = ["backend", "manual_seed"]
