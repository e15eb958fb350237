"""Projected diffusion variants and PointCloudProj (reference diffusion.py:377-429, 525-573,
models.py:75-91): the projection kernel against values computed by the reference, the variants against their base classes."""
import numpy as np
import pytest
import torch

from oracle import oracle as O

DEV = "cuda:0"


def dev(a, dtype=torch.float32):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV).to(dtype)


def test_oracle_projection_vs_reference(golden):
    g = golden["projection"]
    want = np.einsum("pj,bij->bpi", g["data"].astype(np.float64), g["R"].astype(np.float64))   # data @ R^T
    assert np.abs(want - g["proj"]).max() < 2e-6


@pytest.mark.gpu
def test_point_cloud_proj_vs_reference(golden):
    from so3x.models import PointCloudProj
    from so3x.util import euler_to_rmat
    g = golden["projection"]
    proj = PointCloudProj(dev(g["data"]))
    out = proj(dev(g["R"]))
    assert out.shape == (37, 301, 3)
    assert float((out - dev(g["proj"])).abs().max()) < 2e-6
    assert proj(dev(g["R"])[:0]).shape == (0, 301, 3)
    big = proj(dev(g["R"]).repeat(2000, 1, 1))                      # several launches' worth of rotations (grid.y tiling)
    assert torch.equal(big[:37], out) and torch.equal(big[-37:], out)
    eul = dev(g["euler"])
    assert float((euler_to_rmat(*torch.unbind(eul, -1)) - dev(g["euler_rmat"])).abs().max()) < 2e-6
    assert float((PointCloudProj(dev(g["data"]), so3=False)(eul) - dev(g["proj_euler"])).abs().max()) < 5e-6


@pytest.mark.gpu
def test_projected_so3_diffusion_matches_base_with_identity_projection(golden):
    from so3x.diffusion import SO3Diffusion, ProjectedSO3Diffusion
    from so3x.so3_train import RotPredict
    from so3x.models import PointCloudProj
    from so3x import rng
    torch.manual_seed(0)
    net = RotPredict(out_type="skewvec").to(DEV)
    base = SO3Diffusion(net, timesteps=50).to(DEV)
    prj = ProjectedSO3Diffusion(net, timesteps=50).to(DEV)
    x0 = base.p_sample_loop((64,))
    t = torch.randint(0, 50, (64,), device=DEV)
    ident = lambda r: r
    rng.manual_seed(5)
    la = base.p_losses(x0, t)
    rng.manual_seed(5)
    prj.projection = ident
    lb = prj.p_losses(x0, t)
    assert abs(float(la.detach()) - float(lb.detach())) < 1e-6 * abs(float(la.detach()))
    lb.backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in net.parameters())
    # one reverse step with explicit draws: the generic path of the projected class == the fused path of the base class
    gen = torch.Generator(device=DEV).manual_seed(1)
    axes, unif = torch.randn(64, 3, device=DEV, generator=gen), torch.rand(64, device=DEV, generator=gen)
    tt = torch.full((64,), 20, device=DEV, dtype=torch.long)
    a = base.p_sample(x0, tt, axes=axes, unif=unif)
    b = prj.p_sample(x0, tt, axes=axes, unif=unif)
    assert float((a - b).abs().max()) < 2e-5
    # a real projection: the denoiser sees the rotated cloud (here reduced back to 9 numbers so RotPredict can eat it)
    cloud = torch.eye(3, device=DEV)
    proj = PointCloudProj(cloud)                                     # eye @ R^T = R^T
    out = prj.p_sample_loop((32,), lambda r: proj(r).transpose(-1, -2))
    assert out.shape == (32, 3, 3) and torch.isfinite(out).all()
    assert float((out @ out.transpose(-1, -2) - torch.eye(3, device=DEV)).abs().max()) < 1e-4
    loss = prj(x0, lambda r: proj(r).transpose(-1, -2))
    assert torch.isfinite(loss)


@pytest.mark.gpu
def test_projected_se3_diffusion_matches_base_with_identity_projection():
    from so3x.se3 import SE3Diffusion, ProjectedSE3Diffusion, AffineT, AffineGrad
    from so3x import rng, backend as B
    den = lambda x, t: AffineGrad(x.rot[..., 0] * 0.1, x.shift * 0.01)
    base = SE3Diffusion(den, timesteps=40).to(DEV)
    prj = ProjectedSE3Diffusion(den, timesteps=40, shift_scale=75.0).to(DEV)
    n = 128
    x = AffineT(B.quat_to_rmat(torch.randn(n, 4, device=DEV)), torch.randn(n, 3, device=DEV))
    t = torch.randint(0, 40, (n,), device=DEV)
    rng.manual_seed(2)
    la = base.p_losses(x, t)
    rng.manual_seed(2)
    prj.projection = lambda a: a
    lb = prj.p_losses(x, t)
    assert abs(float(la) - float(lb)) < 1e-6 * max(1.0, abs(float(la)))
    out = prj.p_sample_loop((16,), lambda a: a)
    assert out.rot.shape == (16, 3, 3) and torch.isfinite(out.rot).all() and torch.isfinite(out.shift).all()
    assert torch.isfinite(prj(x, lambda a: a))
