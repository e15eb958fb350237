"""SinusoidalPosEmb with the reference's interface (reference models.py:13-25).
On the hot path the embedding is evaluated inside the fused score-network kernels;
this module exists for API/state_dict compatibility and standalone use."""
import math

import torch
from torch import nn

__all__ = ["SinusoidalPosEmb", "Siren", "ResLayer", "PointCloudProj"]


class SinusoidalPosEmb(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.dim = dim

    def forward(self, x):
        half = self.dim // 2
        freqs = torch.exp(torch.arange(half, device=x.device) * -(math.log(10000) / (half - 1)))
        ang = x[:, None] * freqs[None, :]
        return torch.cat((ang.sin(), ang.cos()), dim=-1)


class Siren(nn.Module):
    """Sine-activated positional encoding layer (reference models.py:37-72; imported by so3_train.py:6 and unused on the
    SO(3) path): sin(Linear(x)) with the SIREN initialisation, optionally followed by a Linear.  Plain torch."""

    def __init__(self, in_channels, out_channels, scale=1, optimize=True, post_scale=True):
        super().__init__()
        self.positional = nn.Linear(in_features=in_channels, out_features=out_channels)
        bound = (6 / in_channels) ** 0.5
        nn.init.uniform_(self.positional.weight, -bound, bound)
        self.positional.weight.data *= scale
        nn.init.uniform_(self.positional.bias, -3.14159, 3.14159)  # biases cover +-pi
        self.post_scale = nn.Linear(out_channels, out_channels) if post_scale else None
        for param in self.parameters():
            param.requires_grad = optimize

    def forward(self, x):
        res = torch.sin(self.positional(x))
        return self.post_scale(res) if self.post_scale is not None else res


class ResLayer(nn.Module):
    """x + layer(x) (reference models.py:28-34); the container the wide score network's state_dict keys go through
    (`net.<i>.layer.0.weight`).  so3_lock_train.RotPredict evaluates the whole stack in the fused kernels."""

    def __init__(self, layer: nn.Module):
        super().__init__()
        self.layer = layer

    def forward(self, x):
        return x + self.layer(x)


class PointCloudProj(nn.Module):
    """data @ R^T for a batch of rotations R (reference models.py:75-91): the `projection` callable handed to
    ProjectedSO3Diffusion / ProjectedSE3Diffusion.  so3=False (Euler-angle input of the Euclidean baselines) converts with
    util.euler_to_rmat first."""

    def __init__(self, data, so3=True):
        super().__init__()
        self.data = data
        self.so3 = so3

    def forward(self, x):
        from . import backend as _b
        if not self.so3:
            from .util import euler_to_rmat
            x = euler_to_rmat(*torch.unbind(x, -1))
        return _b.rotate_cloud(x, self.data)
