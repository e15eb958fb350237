"""GPU tests that close the holes the round-1 review listed: the training entry points run end to end on the device (with and
without the captured graph; checkpoints interchange with the reference's key set), BASELINE config 2 at its full size,
the > 2^31-element / grid-cap paths, mixed timesteps in the reverse mean, orthogonalise on non-orthogonal input, the
non-default cosine offset, and the hardware sine / cosine the Philox axis draw uses."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def host(t):
    return t.detach().cpu().numpy()


@pytest.fixture(scope="module")
def mods():
    from so3x import util, distributions, diffusion, so3_train, so3_lock_train, backend, rng
    assert torch.cuda.is_available()
    return dict(util=util, dist=distributions, diff=diffusion, train=so3_train, lock=so3_lock_train, B=backend, rng=rng)


# ------------------------------------------------------------------ BASELINE config 1's workload on the device
@pytest.mark.parametrize("flags", [[], ["--graph"], ["--precision", "bf16"], ["--precision", "bf16", "--graph"],
                                   ["--precision", "bf16", "--graph", "--optimizer", "torch"]])
def test_so3_train_main_runs_and_its_checkpoint_interchanges(mods, golden, tmp_path, capsys, flags):
    """so3x.so3_train.main = the reference's so3_train.py loop (two-mode data, Adam 3e-4) at config 1's size (batch 4096, 100
    diffusion steps): runs eagerly and as a captured graph, fp32 and bf16, logs finite decreasing losses, and saves a
    state_dict with the reference's keys that loads into a fresh RotPredict and drives the sampler."""
    w = tmp_path / "w.pt"
    net = mods["train"].main(["--batch", "4096", "--timesteps", "100", "--steps", "60", "--log-every", "20", "--save-every", "60",
                              "--lr", "3e-3", "--weights", str(w)] + flags)
    lines = [json.loads(l) for l in capsys.readouterr().out.strip().split("\n") if l.startswith("{")]
    assert [l["step"] for l in lines] == [20, 40, 60] and all(np.isfinite(l["loss"]) for l in lines)
    assert lines[-1]["loss"] < lines[0]["loss"]
    sd = torch.load(w, map_location="cpu")
    gkeys = [f"net.{l}.{k}" for l in (0, 2, 4, 6, 8) for k in ("weight", "bias")]
    assert list(sd.keys()) == gkeys                                              # the reference's checkpoint layout
    g = golden["score_mlp"]
    assert all(tuple(sd[k].shape) == g[k.replace(".", "_")].shape for k in gkeys)
    fresh = mods["train"].RotPredict(out_type="skewvec", precision="bf16")
    fresh.load_state_dict(sd)
    fresh = fresh.to(DEV)
    assert torch.equal(fresh.flat_data().cpu(), net.flat_data().cpu())
    proc = mods["diff"].SO3Diffusion(fresh, timesteps=100).to(DEV)
    x = proc.p_sample_loop((512,))
    assert torch.isfinite(x).all() and float((x @ x.transpose(-1, -2) - torch.eye(3, device=DEV)).abs().max()) < 1e-5
    # and the other way round: the reference-initialised golden weights load into the trained module
    net.load_state_dict({k: torch.from_numpy(g[k.replace(".", "_")]) for k in gkeys})
    assert net._flat_ok()


def test_so3_lock_train_main_runs_with_graph(mods, tmp_path, capsys):
    w = tmp_path / "wl.pt"
    net = mods["lock"].main(["--batch", "256", "--timesteps", "100", "--steps", "12", "--log-every", "4", "--save-every", "12",
                             "--precision", "bf16", "--graph", "--weights", str(w)])
    lines = [json.loads(l) for l in capsys.readouterr().out.strip().split("\n") if l.startswith("{")]
    assert len(lines) == 3 and all(np.isfinite(l["loss"]) for l in lines)
    sd = torch.load(w, map_location="cpu")
    assert list(sd.keys())[0] == "net.0.layer.0.weight" and sum(v.numel() for v in sd.values()) == 392448
    assert net._flat_ok()


# ------------------------------------------------------------------ BASELINE config 2 at its full size
def test_config_2_density_and_score_at_2p20(mods):
    """IGSO(3) log-density + score for 2^20 rotations with per-sample eps (config 2b) and scalar eps (2a): finite where the
    reference is, score parallel to the rotation axis (= (f'/f) axis: orthogonal to nothing else), the dense gradient is the
    tangent form pushed through d omega / dR, and a 4096-sample subset equals the f64 oracle."""
    B = mods["B"]
    n = 1 << 20
    g = torch.Generator(device=DEV).manual_seed(0)
    R = B.quat_to_rmat(torch.randn(n, 4, device=DEV, generator=g))
    tt = torch.randint(0, 1000, (n,), device=DEV, generator=g)
    sched = torch.from_numpy(B.schedule_from_betas(B.cosine_beta_schedule(1000))).to(DEV)
    eps = sched[4][tt].contiguous()                                   # eps_i = sqrt(1 - abar_{t_i})
    for e in (eps, torch.tensor(0.5, device=DEV)):
        logp, score, grad = B.igso3_logprob_score(R, e, want_score=True, want_grad=True)
        axis, ang = B.rmat_to_aa(R)
        ev = e if e.numel() > 1 else e.expand(n)
        # the reference's density underflows to 0 (log -> -inf) beyond its overflow cut-off; everywhere else it is finite
        fin = torch.isfinite(logp[:, 0])
        assert float(fin.float().mean()) > 0.5 if e.numel() > 1 else bool(fin.all())
        assert torch.isfinite(score[fin]).all() and torch.isfinite(grad[fin]).all()
        cross = torch.linalg.cross(score[fin], axis[fin])
        assert float(cross.abs().max()) < 2e-4 * max(1.0, float(score[fin].abs().max()))          # score || axis
        sub = torch.randperm(n, device=DEV, generator=g)[:4096]
        ref = O.igso3_log_prob(host(R[sub]), host(ev[sub]))
        got = host(logp[sub, 0])
        # The finite / -inf pattern.  The reference's log-density is -inf in exactly two cases
        # (distributions.py:53-77): (i) (w - 2 pi) exp(pi w / v) overflows float64 and `vals[vals.isinf()] = 0` zeroes the density:
        # x + ln(2 pi - w) > ln(DBL_MAX) = 709.78 with x = pi w / v; (ii) the float64 value is below half the smallest fp32
        # denormal and `.float()` rounds it to 0: ln f < ln 2^-150 = -103.97.  Both conditions read the angle w multiplied by
        # up to pi / v = 8e4 (eps_0 = 6e-3), and the angle read off an fp32 rotation matrix carries ~3e-7 whatever the formula
        # (device: atan2 in fp32; checker: float64 arithmetic on the same fp32 entries): the two sides of either threshold can
        # differ by ~0.03 in the exponent.  So: the pattern must be IDENTICAL wherever both margins exceed 0.1, and such
        # samples must be all but a sliver of the batch; only inside the band may a sample fall on the other side.
        _, w64 = O.rmat_to_aa(host(R[sub]), "f64")
        w64 = w64[:, 0].astype(np.float64)
        v64 = host(ev[sub]).astype(np.float64) ** 2
        x64 = np.pi * w64 / v64
        m_over = x64 + np.log(2 * np.pi - w64) - 709.782712893384
        e1, e2 = np.exp(-np.pi * (np.pi - w64) / v64), np.exp(-np.pi * (np.pi + w64) / v64)
        g64 = w64 - (w64 - 2 * np.pi) * e1 - (w64 + 2 * np.pi) * e2
        with np.errstate(divide="ignore", invalid="ignore"):
            lnf = 0.5 * np.log(np.pi) - 1.5 * np.log(v64) + v64 / 4 - w64 ** 2 / (4 * v64) + np.log(g64) - np.log(2 * np.sin(w64 / 2))
        m_under = lnf + 150 * np.log(2.0)
        decided = (np.abs(m_over) > 0.1) & (np.abs(m_under) > 0.1) & (w64 > 0)
        assert decided.mean() > 0.99, decided.mean()
        want_finite = (m_over < 0) & (m_under > 0)
        assert (np.isfinite(ref)[decided] == want_finite[decided]).all()    # the checker agrees with the two stated conditions
        assert (np.isfinite(got)[decided] == np.isfinite(ref)[decided]).all(), int((np.isfinite(got) != np.isfinite(ref))[decided].sum())
        assert (np.isfinite(ref) == np.isfinite(got)).mean() > 0.995        # and inside the band only a handful may flip
        both = np.isfinite(ref) & np.isfinite(got)
        assert both.sum() > 1000
        assert np.abs(got[both] - ref[both]).max() < 2e-4 * np.maximum(1.0, np.abs(ref[both])).max()
        dl = O.igso3_dlogf(host(ang[sub, 0]), host(ev[sub]))                                        # f'/f in f64
        sc = host((score[sub] * axis[sub]).sum(-1))                                                 # its component along the axis
        okm = both & (host(ang[sub, 0]) > 1e-2) & (host(ang[sub, 0]) < 3.1)
        assert np.abs(sc[okm] - dl[okm]).max() < 2e-3 * np.maximum(1.0, np.abs(dl[okm])).max()


@pytest.mark.slow
def test_big_batches_beyond_2p31_elements(mods):
    """2^28 + 12,345 rotations (element indices beyond 2^31, grids beyond the 2^20-block cap) through the streaming kernels,
    the noising kernel and two chain steps: the tail of every result equals the same call on the tail alone"""
    B = mods["B"]
    free, _ = torch.cuda.mem_get_info()
    if free < 60 << 30:
        pytest.skip("needs ~50 GB of free device memory")
    n = (1 << 28) + 12345
    R = B.quat_to_rmat(torch.randn(n, 4, device=DEV))
    tail = slice(n - 1000, n)
    k = torch.rand(n, device=DEV)
    assert torch.equal(B.so3_scale(R, k)[tail], B.so3_scale(R[tail].contiguous(), k[tail].contiguous()))
    e = k * 0.9 + 0.1
    lp, sc, _ = B.igso3_logprob_score(R, e)
    lp2, sc2, _ = B.igso3_logprob_score(R[tail].contiguous(), e[tail].contiguous())
    assert torch.equal(lp[tail], lp2) and torch.equal(sc[tail], sc2)
    del lp, sc
    net = mods["train"].RotPredict(out_type="skewvec", precision="bf16").to(DEV)
    proc = mods["diff"].SO3Diffusion(net, timesteps=1000).to(DEV)
    tq, tp = proc._tables()
    t = torch.randint(0, 1000, (n,), device=DEV)
    xt, tg, _ = B.q_sample_target(proc._sched, tq, R, t, seed=1, rng_offset=3, guide_q=proc._guide_q, quirk_col0=False)
    xt2, tg2, _ = B.q_sample_target(proc._sched, tq, R[tail].contiguous(), t[tail].contiguous(), seed=1, rng_offset=3,
                                    index_base=n - 1000, guide_q=proc._guide_q, quirk_col0=False)
    assert torch.equal(xt[tail], xt2) and torch.equal(tg[tail], tg2)
    del xt, tg
    out = B.p_sample_chain(net.flat_data(), proc._sched, tp, R, 500, 2, seed=2, precision=1, guide_p=proc._guide_p)
    out2 = B.p_sample_chain(net.flat_data(), proc._sched, tp, R[tail].contiguous(), 500, 2, seed=2, precision=1, index_base=n - 1000,
                            guide_p=proc._guide_p)
    assert torch.equal(out[tail], out2) and torch.isfinite(out).all()


# ------------------------------------------------------------------ per-sample timesteps in the reverse mean
def test_mixed_timesteps_use_each_samples_coefficients(mods, golden):
    """predict_start_from_noise / p_mean_variance / p_sample with a batch of DIFFERENT timesteps gather the schedule
    coefficients per sample, as the reference's extract(coef, t, shape) does (diffusion.py:291-313) -- round 1 silently used
    t[0] for all; p_sample keeps the reference's noise rule (scale of t[0], none only when every t is 0)"""
    B = mods["B"]
    torch.manual_seed(0)
    net = mods["train"].RotPredict(out_type="skewvec").to(DEV)
    T = 100
    proc = mods["diff"].SO3Diffusion(net, timesteps=T).to(DEV)
    n = 300
    x = B.quat_to_rmat(torch.randn(n, 4, device=DEV))
    v = torch.randn(n, 3, device=DEV) * 0.3
    t = torch.randint(0, T, (n,), device=DEV)
    sched = O.schedule_from_betas(O.cosine_beta_schedule(T))
    x0h = proc.predict_start_from_noise(x, t, v)
    ref_x0h = np.stack([O.p_mean(host(x[i:i + 1]), host(v[i:i + 1]), *(float(sched[r][int(t[i])]) for r in (6, 7, 10, 11)), "f64")[0][0]
                        for i in range(n)])
    well = O.rmat_to_aa(host(x), "f64")[1][:, 0] < 3.0
    # exp(a log x) amplifies the fp32 rounding of log x by a = sqrt(1 / abar_t), which reaches ~1e3 here
    a_i = sched[6][host(t)]
    err = np.abs(host(x0h) - ref_x0h).reshape(n, -1).max(1)
    assert (err[well] < np.maximum(2e-5, 2e-6 * a_i[well])).all() and np.median(err) < 2e-6
    assert np.abs(host(proc.predict_start_from_noise(x, t[:1], v)) -                                    # (1,)-shaped t: shared
                  host(proc.predict_start_from_noise(x, torch.full((n,), int(t[0]), device=DEV), v))).max() == 0
    mean, var, logvar = proc.p_mean_variance(x, t)
    vnet = net(x, t)
    ref_mean = np.stack([O.p_mean(host(x[i:i + 1]), host(vnet[i:i + 1]), *(float(sched[r][int(t[i])]) for r in (6, 7, 10, 11)), "f64")[1][0]
                         for i in range(n)])
    well2 = well & (O.rmat_to_aa(ref_x0h if False else host(x0h), "f64")[1][:, 0] < 3.0)
    assert np.median(np.abs(host(mean) - ref_mean).reshape(n, -1).max(1)) < 2e-6
    assert var.shape == (n,) and torch.equal(var, proc.posterior_variance[t])
    # p_sample with mixed t: mean per sample, noise of sigma[t[0]] (explicit draws make it reproducible)
    ax, un = torch.randn(n, 3, device=DEV), torch.rand(n, device=DEV)
    out = proc.p_sample(x, t, axes=ax, unif=un)
    _, trap_p = proc._tables()
    smp, _, _ = B.igso3_sample(trap_p, n, row_const=int(t[0]), axes=ax, unif=un)
    assert float((out - B.rmul(mean, smp)).abs().max()) < 1e-6
    same = proc.p_sample(x, torch.full((n,), 7, device=DEV), axes=ax, unif=un)
    assert torch.equal(same, proc.p_sample(x, 7, axes=ax, unif=un))                                   # all-equal tensor == int


def test_orthogonalise_does_what_the_reference_does(mods, golden):
    g = golden["orthogonalise"]
    util = mods["util"]
    for k in ("pert", "general", "affine"):
        got = host(util.orthogonalise(torch.from_numpy(g[k + "_in"]).to(DEV)))
        assert np.abs(got - g[k + "_out"]).max() < 5e-6, k
    r = util.quat_to_rmat(torch.randn(1000, 4, device=DEV))
    assert float((util.orthogonalise(r) - r).abs().max()) < 2e-6       # the identity map on rotations (to rounding)


def test_cosine_schedule_with_another_offset(mods):
    """cosine_beta_schedule(T, s) for s != 0.008 evaluates the published formula instead of refusing"""
    d = mods["diff"]
    b = d.cosine_beta_schedule(50, s=0.02)
    x = np.linspace(0, 51, 51)
    ac = np.cos(((x / 51) + 0.02) / 1.02 * np.pi * 0.5) ** 2
    assert np.allclose(b, np.clip(1 - (ac[1:] / ac[0]) / (ac[:-1] / ac[0]), 0, 0.999), rtol=1e-12)
    assert np.array_equal(d.cosine_beta_schedule(50), mods["B"].cosine_beta_schedule(50))
    proc = d.SO3Diffusion(mods["train"].RotPredict(out_type="skewvec"), betas=b)
    assert proc.num_timesteps == 50


def test_philox_axes_are_unit_vectors_uniform_on_the_sphere(mods):
    """the noise axis comes from the hardware sine / cosine (so3x_math.hpp sincos_rev): unit length to 2e-6, mean direction 0,
    second moments 1/3 -- the distribution of the reference's normalised Gaussian 3-vector (distributions.py:35-36)"""
    B = mods["B"]
    trap = B.igso3_build_tables(torch.tensor([0.5], device=DEV))
    n = 1 << 20
    _, ang, ax = B.igso3_sample(trap, n, row_const=0, seed=3, rng_offset=1, want_angle=True, want_axis=True)
    nrm = ax.norm(dim=-1)
    assert float((nrm - 1).abs().max()) < 2e-6
    assert float(ax.mean(0).abs().max()) < 4e-3 and float(((ax ** 2).mean(0) - 1 / 3).abs().max()) < 3e-3
    az = torch.atan2(ax[:, 1], ax[:, 0])                                  # azimuth uniform on (-pi, pi]
    hist = torch.histc(az, bins=64, min=-np.pi, max=np.pi) / n * 64
    assert float((hist - 1).abs().max()) < 0.03
