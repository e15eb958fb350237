"""The reference's data source and evaluation harness around the hot path (bingham_train.py / bingham_test.py,
distributions.py:113-127): Bingham-distributed quaternions -> rotations, and the MMD between them and the diffusion's
samples.  CPU: the distribution and the module interface.  GPU: the evaluation pipeline end to end."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import PKG

DEV = "cuda:0"


def test_bingham_is_a_normalised_zero_mean_gaussian():
    from so3x.distributions import Bingham
    from so3x.bingham_train import covpairs, loc, BATCH, RotPredict
    assert BATCH == 64 and [a for _, a, _ in covpairs] == ["sur", "scr", "lcr", "lur"]
    torch.manual_seed(0)
    for _, acro, cov in covpairs:
        assert cov.shape == (4, 4)
        d = Bingham(loc + 3.0, covariance_matrix=cov)          # the location argument is ignored (always zero)
        assert torch.equal(d.loc, torch.zeros(4))
        s = d.sample((20000,))
        assert s.shape == (20000, 4) and float((s.norm(dim=-1) - 1).abs().max()) < 1e-6
        assert float(s.mean(0).abs().max()) < 0.03               # antipodally symmetric
        # the unnormalised draw has the requested covariance: compare second moments of the directions with a direct draw
        g = torch.distributions.MultivariateNormal(torch.zeros(4), covariance_matrix=cov).sample((20000,))
        g = g / g.norm(dim=-1, keepdim=True)
        assert float(((s.T @ s) / 20000 - (g.T @ g) / 20000).abs().max()) < 0.03
    # small rotations <-> real part near +-1; uniform rotations <-> E|w| = 8 / (3 pi) / 2
    assert float(Bingham(loc, covariance_matrix=covpairs[0][2]).sample((5000,))[:, 0].abs().mean()) > 0.98
    assert abs(float(Bingham(loc, covariance_matrix=covpairs[3][2]).sample((50000,))[:, 0].abs().mean()) - 8 / (3 * np.pi) / 2) < 0.01
    assert RotPredict(out_type="skewvec").flat_params().numel() == 17358


def test_bingham_flat_name_shims(tmp_path):
    code = ("from bingham_train import covpairs, RotPredict, loc\n"
            "from bingham_test import calc_step, SAMPLES, NET_SAMPLES\n"
            "from distributions import Bingham, IsotropicGaussianSO3\n"
            "assert SAMPLES == 20000 and callable(calc_step) and len(covpairs) == 4\n"
            "print('OK')\n")
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([os.path.join(PKG, "compat"), PKG]))
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, cwd=str(tmp_path), timeout=180)
    assert r.returncode == 0 and "OK" in r.stdout, r.stderr


@pytest.mark.gpu
def test_bingham_mmd_separates_the_settings():
    """two independent draws of one setting pass the reference's kernel two-sample test; different settings do not"""
    from so3x.distributions import Bingham
    from so3x.bingham_train import covpairs, loc
    from so3x import util
    torch.manual_seed(0)
    n = 8192
    draws = {}
    for _, acro, cov in covpairs:
        d = Bingham(loc.to(DEV), covariance_matrix=cov.to(DEV))
        draws[acro] = (util.quat_to_rmat(d.sample((n,))), util.quat_to_rmat(d.sample((n,))))
    for acro, (a, b) in draws.items():
        assert util.Ker_2samp_test(a, b, util.rmat_gaussian_kernel), acro
    assert not util.Ker_2samp_test(draws["sur"][0], draws["lur"][0], util.rmat_gaussian_kernel)
    assert not util.Ker_2samp_test(draws["sur"][0], draws["lcr"][0], util.rmat_gaussian_kernel)
    assert float(util.MMD(draws["sur"][0], draws["lur"][0], util.rmat_gaussian_kernel)) > \
        10 * float(util.MMD(*draws["lur"], util.rmat_gaussian_kernel))


@pytest.mark.gpu
def test_bingham_train_then_evaluate_pipeline(tmp_path):
    """bingham_train.main for a few hundred steps on one setting, then bingham_test.calc_step on the saved weights: the
    MMD of the briefly trained model is finite and already far below that of an untrained one."""
    from so3x import bingham_train, bingham_test
    import so3x
    so3x.manual_seed(3)
    torch.manual_seed(3)
    wd = str(tmp_path / "weights")
    bingham_train.main(["--steps", "600", "--batch", "4096", "--timesteps", "100", "--precision", "bf16", "--lr", "2e-3",
                        "--save-every", "600", "--weights-dir", wd, "--cov", "sur"])
    assert os.path.exists(os.path.join(wd, "weights_bing_sur_600.pt")) and os.path.exists(os.path.join(wd, "weights_bing_sur_0.pt"))
    cov, = [c for _, a, c in bingham_train.covpairs if a == "sur"]
    kw = dict(weights_dir=wd, samples=4096, net_samples=4096, timesteps=100, precision="bf16")
    trained = bingham_test.calc_step("sur", cov, 600, **kw)
    untrained = bingham_test.calc_step("sur", cov, 0, **kw)
    assert np.isfinite(trained) and np.isfinite(untrained)
    assert trained < 0.25 * untrained


# ----------------------------------------------------------------------------- the sampling scripts (so3_test.py, so3_lock_test.py)
def test_rmat_to_euler_inverts_euler_to_rmat():
    from so3x import util
    g = torch.Generator().manual_seed(5)
    x = (torch.rand(200, generator=g) - 0.5) * 6.0
    y = (torch.rand(200, generator=g) - 0.5) * 3.0          # |y| < pi/2: the decomposition's principal range
    z = (torch.rand(200, generator=g) - 0.5) * 6.0
    xr, yr, zr = util.rmat_to_euler(util.euler_to_rmat(x, y, z))
    for a, b in ((x, xr), (y, yr), (z, zr)):
        assert float((a - b).abs().max()) < 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("wide", [False, True])
def test_sampling_script_trajectory(tmp_path, wide):
    """so3_test.py / so3_lock_test.py: weights file -> external per-step loop with the whole trajectory kept"""
    import so3x
    from so3x import so3_test, so3_lock_test, util
    from so3x.diffusion import SO3Diffusion
    if wide:
        from so3x.so3_lock_train import RotPredict
    else:
        from so3x.so3_train import RotPredict
    torch.manual_seed(1)
    net = RotPredict(out_type="skewvec")
    wpath = str(tmp_path / "w.pt")
    torch.save(net.state_dict(), wpath)
    proc = SO3Diffusion(net.to(DEV), timesteps=50).to(DEV)
    R0 = util.quat_to_rmat(torch.randn(96, 4, device=DEV))
    so3x.manual_seed(11)
    res, final = so3_test.sample_trajectory(proc, R0)
    assert res.shape == (50, 96, 3, 3) and torch.equal(res[49], R0)
    eye = torch.eye(3, device=DEV)
    assert float((res @ res.transpose(-1, -2) - eye).abs().max()) < 2e-5 and torch.isfinite(final).all()
    so3x.manual_seed(11)                                    # same seed, same launch sequence: identical trajectory
    res2, final2 = so3_test.sample_trajectory(proc, R0)
    assert torch.equal(res, res2) and torch.equal(final, final2)
    d = so3_test.mode_distance(res)
    assert d.shape == (50, 96) and float(d.min()) >= 0 and float(d.max()) <= np.pi + 1e-4
    z90 = torch.tensor([[0.0, -1.0, 0.0], [1.0, 0.0, 0.0], [0.0, 0.0, 1.0]], device=DEV)
    both = torch.stack([z90, z90.T])[None].repeat(3, 1, 1, 1)
    assert float(so3_test.mode_distance(both).abs().max()) < 1e-3   # the two modes themselves are at distance 0
    main = so3_lock_test.main if wide else so3_test.main
    out = str(tmp_path / "traj.pt")
    s = main(["--weights", wpath, "--timesteps", "20", "--batch", "40", "--out", out])
    assert s["timesteps"] == 20 and torch.load(out)["trajectory"].shape == (20, 40, 3, 3)
