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


# ------------------------------------------------------------------ PlaneNet (reference models.py:185-210)
def _planenet(golden, device="cpu", dropout=0.0):
    from so3x.models import PlaneNet
    g = golden["planenet"]
    net = PlaneNet(dim=int(g["dim"]), heads=int(g["heads"]), layers=int(g["layers"]), dropout=dropout).eval()
    sd = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd_")}
    assert set(net.state_dict().keys()) == set(sd.keys())                      # the reference's checkpoint keys
    net.load_state_dict(sd)
    return net.to(device), g


def test_planenet_blocks_vs_reference(golden):
    """PlaneNet with the reference's weights reproduces the reference's own submodules run in its own order (SIREN + time
    embedding -> TransformerEncoder -> PoolRN with an all-true [B, P] mask -> Linear): CPU, torch against torch"""
    net, g = _planenet(golden)
    with torch.no_grad():
        out = net.forward_torch(torch.from_numpy(g["x"]), torch.from_numpy(g["t"]))
        x_emb = net.position_siren(torch.from_numpy(g["x"]))
        t_in = torch.cat((x_emb, net.time_embedding(torch.from_numpy(g["t"]))[:, None, :].expand(x_emb.shape)), dim=2)
        enc = net.encoder(t_in.transpose(0, 1)).transpose(0, 1)
    assert out.shape == (6, 3)
    assert np.abs(enc.numpy() - g["encoding"]).max() < 1e-5
    assert np.abs(out.numpy() - g["out"]).max() < 1e-5
    # a masked pool ignores the masked points
    from so3x.models import PoolRN
    torch.manual_seed(0)
    pool = PoolRN(8)
    x = torch.randn(3, 5, 8)
    m = torch.tensor([[1, 1, 1, 0, 0]] * 3, dtype=torch.bool)
    assert torch.allclose(pool(x, m), pool(x[:, :3]), atol=1e-6)


@pytest.mark.gpu
def test_planenet_as_the_denoiser_of_projected_so3_diffusion(golden):
    """the aircraft task's wiring (aircraft_rotate.py:64-106): a batch of point clouds, one pose each, PointCloudProj as the
    projection, PlaneNet as the denoiser of ProjectedSO3Diffusion -- the noising / target / posterior / noise steps AND the
    transformer (so3x_planenet_fwd / _bwd, through autograd) are this package's kernels.  The network equals the reference's on the GPU too, per-sample clouds equal
    torch.matmul's batching, a few Adam steps lower the loss, and the reverse loop returns rotations.  The network is built and trained
    as the reference does it: torch's default dropout 0.1, `net.train()` (aircraft_rotate.py:66) -- the kernels' own dropout."""
    from so3x.diffusion import ProjectedSO3Diffusion
    from so3x.models import PointCloudProj
    from so3x import backend as B
    net, g = _planenet(golden, DEV, dropout=0.1)
    with torch.no_grad():
        out = net(dev(g["x"]), dev(g["t"], torch.int64))
    assert float((out - dev(g["out"])).abs().max()) < 1e-4
    bsz, pts = 16, 24
    gen = torch.Generator(device=DEV).manual_seed(3)
    clouds = torch.randn(bsz, pts, 3, device=DEV, generator=gen) * 0.5
    R = B.quat_to_rmat(torch.randn(bsz, 4, device=DEV, generator=gen))
    proj = PointCloudProj(clouds)
    assert float((proj(R) - clouds @ R.transpose(-1, -2)).abs().max()) < 2e-6    # one cloud per rotation
    net.train()
    process = ProjectedSO3Diffusion(net, timesteps=50).to(DEV)
    truepos = process.identity.repeat(bsz, 1, 1)
    opt = torch.optim.Adam(net.parameters(), lr=1e-3)
    torch.manual_seed(0)
    assert torch.isfinite(process(truepos, proj))                                  # the reference's call, fresh t and noise
    # ... and on a FIXED batch of timesteps and draws the loss must go down (fresh noise every step hides that in 30 steps)
    t = torch.randint(0, 50, (bsz,), device=DEV, generator=gen)
    ax, un = torch.randn(bsz, 3, device=DEV, generator=gen), torch.rand(bsz, device=DEV, generator=gen)
    process.projection = proj
    losses = []
    for _ in range(30):
        loss = process.p_losses(truepos, t, axes=ax, unif=un)
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    assert all(np.isfinite(losses)) and min(losses[-5:]) < 0.7 * losses[0]      # (a fresh dropout mask every step)
    net.eval()
    x = process.p_sample_loop((bsz,), proj)
    assert x.shape == (bsz, 3, 3) and torch.isfinite(x).all()
    assert float((x @ x.transpose(-1, -2) - torch.eye(3, device=DEV)).abs().max()) < 1e-4
