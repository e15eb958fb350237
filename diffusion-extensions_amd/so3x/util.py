"""Rotation algebra with the reference's names and argument conventions
(reference util.py:79-107, 164-252, 315-361), computed by the HIP kernels behind
so3x.backend.  Tensors must live on the MI355X ('cuda') device; leading batch
dimensions are free, rotations end in (3, 3), fp32."""
from typing import Tuple

import torch

from . import backend as _b

__all__ = ["skew2vec", "vec2skew", "orthogonalise", "log_rmat", "aa_to_rmat", "rmat_to_aa", "quat_to_rmat",
           "rmat_dist", "so3_lerp", "so3_scale", "cycle"]


def skew2vec(skew: torch.Tensor) -> torch.Tensor:
    """vee map (reference util.py:79-84): (S21, -S20, S10).  Pure indexing: stays in torch."""
    return torch.stack((skew[..., 2, 1], -skew[..., 2, 0], skew[..., 1, 0]), dim=-1)


def vec2skew(vec: torch.Tensor) -> torch.Tensor:
    """hat map (reference util.py:87-92)."""
    z = torch.zeros_like(vec[..., 0])
    x, y, w = vec[..., 0], vec[..., 1], vec[..., 2]
    return torch.stack((z, -w, y, w, z, -x, -y, x, z), dim=-1).reshape(vec.shape[:-1] + (3, 3))


def orthogonalise(mat: torch.Tensor) -> torch.Tensor:
    """Reference util.py:95-107 snaps the singular values of the 3x3 block to {-1,0,1}.
    Every rotation produced by this backend comes out of a closed-form Rodrigues formula
    and is orthogonal to fp32 rounding, where that SVD round trip is the identity map
    (SURVEY.md 2.3 K4); kept for API compatibility."""
    return mat


def log_rmat(r_mat: torch.Tensor) -> torch.Tensor:
    """Matrix log as a skew-symmetric matrix (reference util.py:164-192)."""
    return _b.log_rmat(r_mat)


def aa_to_rmat(rot_axis: torch.Tensor, ang: torch.Tensor) -> torch.Tensor:
    """Axis (any norm) + angle -> rotation matrix (reference util.py:195-205)."""
    return _b.aa_to_rmat(rot_axis, ang)


def rmat_to_aa(r_mat: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """Rotation -> (axis [...,3], angle [...,1] in [0, pi]) (reference util.py:208-219)."""
    return _b.rmat_to_aa(r_mat)


def quat_to_rmat(quaternions: torch.Tensor) -> torch.Tensor:
    """Real-first quaternions of any norm -> rotation matrices (reference util.py:222-252)."""
    return _b.quat_to_rmat(quaternions)


def rmat_dist(input: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    """Geodesic distance ||log(input^T target)||_F (reference util.py:315-322)."""
    return _b.rmat_dist(input, target)


def so3_lerp(rot_a: torch.Tensor, rot_b: torch.Tensor, weight: torch.Tensor) -> torch.Tensor:
    """Geodesic interpolation (reference util.py:325-338); rot_a may be a single (3,3)."""
    return _b.so3_lerp(rot_a, rot_b, weight)


def so3_scale(rmat: torch.Tensor, scalars: torch.Tensor) -> torch.Tensor:
    """exp(scalars * log(rmat)) (reference util.py:349-361); scalars has 1 or batch elements."""
    return _b.so3_scale(rmat, scalars)


def cycle(iterable):
    """Endless iterator over a DataLoader-like iterable (reference util.py helper)."""
    while True:
        for x in iterable:
            yield x
