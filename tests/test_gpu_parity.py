"""GPU parity tests (run with -m gpu on an MI355X).  Every check goes through the
Python host layer -> ctypes -> the C ABI of libso3x.so and compares against the CPU
oracle (oracle/) and the golden vectors captured from the reference.
Gates G1-G5: SURVEY.md section 8c."""
import os

import numpy as np
import pytest
import torch

from oracle import oracle as O

pytestmark = pytest.mark.gpu


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]

DEV = "cuda:0"


def dev(a, dtype=torch.float32):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV).to(dtype)


def host(t):
    return t.detach().cpu().numpy()


def maxabs(a, b):
    return float(np.max(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64))))


def frob_err(a, b):
    n = a.shape[0]
    return np.linalg.norm((np.asarray(a, np.float64) - np.asarray(b, np.float64)).reshape(n, -1), axis=1) / np.sqrt(3)


@pytest.fixture(scope="module")
def mods():
    from so3x import util, distributions, diffusion, so3_train, backend
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return dict(util=util, dist=distributions, diff=diffusion, train=so3_train, B=backend)


@pytest.fixture(scope="module")
def net(mods, golden):
    g = golden["score_mlp"]
    n = mods["train"].RotPredict(out_type="skewvec")
    n.load_state_dict({f"net.{l}.{k}": torch.from_numpy(g[f"net_{l}_{k}"]) for l in (0, 2, 4, 6, 8) for k in ("weight", "bias")})
    return n.to(DEV)


# ------------------------------------------------------------------ rotation algebra (G1)
def test_rotation_ops_vs_golden_and_oracle(mods, golden):
    U = mods["util"]
    g = golden["rotation_ops"]
    q, R, R2 = dev(g["q"]), dev(g["R_32"]), dev(g["R2_32"])
    assert maxabs(host(U.quat_to_rmat(q)), g["R_32"]) < 1e-6
    assert maxabs(host(U.log_rmat(R)), g["log_64"]) < 1e-5
    assert maxabs(host(U.so3_scale(R, dev(g["k"]))), g["scale_64"]) < 1e-5
    assert maxabs(host(U.aa_to_rmat(dev(g["axis_in"]), dev(g["ang_in"]))), g["aa2r_64"]) < 1e-5
    ax, an = U.rmat_to_aa(R)
    assert maxabs(host(ax), g["r2aa_axis_64"]) < 1e-5 and maxabs(host(an), g["r2aa_angle_64"]) < 1e-5
    assert an.shape == (64, 1)
    assert maxabs(host(U.so3_lerp(R, R2, dev(g["w"]))), g["lerp_64"]) < 1e-5
    assert maxabs(host(U.rmat_dist(R, R2)), g["dist_64"]) < 1e-5
    assert torch.all(U.log_rmat(torch.eye(3, device=DEV)[None]) == 0)
    # scalar broadcast forms used by diffusion.py (t of shape (1,))
    assert maxabs(host(U.so3_scale(R, torch.tensor([0.37], device=DEV))), O.so3_scale(g["R_32"], np.float64(np.float32(0.37)), "f64")) < 1e-5
    eye = torch.eye(3, device=DEV)
    assert maxabs(host(U.so3_lerp(eye, R2, dev(g["w"]))), O.so3_lerp(np.eye(3)[None], g["R2_32"], g["w"], "f64")) < 1e-5


@pytest.mark.parametrize("n", [1, 63, 255, 256, 257, 4099])
def test_rotation_ops_ragged_sizes(mods, n):
    U = mods["util"]
    rng = np.random.default_rng(n)
    q = rng.standard_normal((n, 4)).astype(np.float32)
    k = rng.uniform(-2, 2, n).astype(np.float32)
    R = U.quat_to_rmat(dev(q))
    assert maxabs(host(R), O.quat_to_rmat(q, "f64")) < 1e-6
    Rh = host(R)
    assert maxabs(host(U.so3_scale(R, dev(k))), O.so3_scale(Rh, k, "f64")) < 1e-5
    assert maxabs(host(mods["B"].log_rmat_vec(R)), O.log_rmat_vec(Rh, "f64")) < 1e-5
    w = rng.standard_normal((n, 3)).astype(np.float32)
    assert maxabs(host(mods["B"].exp_skewvec(dev(w))), O.exp_vec(w, "f64")) < 1e-5


def test_empty_batches(mods):
    U = mods["util"]
    assert U.quat_to_rmat(torch.empty(0, 4, device=DEV)).shape == (0, 3, 3)
    assert U.so3_scale(torch.empty(0, 3, 3, device=DEV), torch.empty(0, device=DEV)).shape == (0, 3, 3)


def test_rotation_properties_full_size(mods):
    """BASELINE size (2^20): size-independent invariants instead of a slow CPU comparison."""
    U = mods["util"]
    n = 1 << 20
    g = torch.Generator(device=DEV).manual_seed(0)
    R = U.quat_to_rmat(torch.randn(n, 4, device=DEV, generator=g))
    RRt = R @ R.transpose(-1, -2)
    assert float((RRt - torch.eye(3, device=DEV)).abs().max()) < 5e-6
    assert float((torch.linalg.det(R) - 1).abs().max()) < 5e-6
    back = mods["B"].exp_skewvec(U.skew2vec(U.log_rmat(R)))
    err = (back - R).abs().amax(dim=(1, 2))
    _, ang = U.rmat_to_aa(R)
    well = ang[:, 0] < 3.0   # log is ill-conditioned like 1e-7/(pi - omega)
    assert float(err[well].max()) < 1e-5
    assert float((U.so3_scale(R, torch.ones(n, device=DEV)) - R).abs().amax(dim=(1, 2))[well].max()) < 1e-5
    assert float(U.rmat_dist(R, R).max()) < 2e-3      # sqrt(2)*angle of a near-identity product, fp32 noise floor
    R2 = U.quat_to_rmat(torch.randn(n, 4, device=DEV, generator=g))
    assert float((U.so3_lerp(R, R2, torch.zeros(n, device=DEV)) - R).abs().max()) < 1e-5
    d = U.rmat_dist(R, R2)
    _, a2 = U.rmat_to_aa(R.transpose(-1, -2) @ R2)
    assert float((d - a2[:, 0] * 2 ** 0.5).abs().max()) < 1e-4


# ------------------------------------------------------------------ IGSO(3) A1-A3
def test_eps_ft_vs_golden(mods, golden):
    g = golden["eps_ft"]
    om = dev(g["omega"])
    for j, e in enumerate(g["eps"]):
        d = mods["dist"].IsotropicGaussianSO3(torch.tensor(float(e), device=DEV))
        mine = host(d._eps_ft(om))
        ref = g["vals"][:, j]
        ok = (np.isnan(mine) & np.isnan(ref)) | (mine == ref) | (np.abs(mine - ref) <= 2e-6 * np.abs(ref))
        assert ok.all(), (e, mine[~ok], ref[~ok])


def test_tables_vs_golden_G4(mods, golden):
    g = golden["igso3_tables"]
    trap = host(mods["B"].igso3_build_tables(dev(g["eps"])))
    assert maxabs(trap, g["trap"]) <= 1e-6
    assert (np.diff(trap, axis=1) >= 0).all() and (trap[:, -1] == 1).all() and (trap[:, :43] == 0).all()
    # all 2 x 1000 schedule rows against the oracle
    sched = O.schedule_from_betas(O.cosine_beta_schedule(1000))
    for row in (4, 9):
        eps = sched[4] if row == 4 else np.exp(np.float32(0.5) * sched[9])
        mine = host(mods["B"].igso3_build_tables(dev(eps)))
        ref = O.igso3_build_tables(eps)
        # sigma_0 = 1e-10 gives an all-NaN row on both sides (0/0 normalisation; never sampled: t = 0 adds no noise)
        assert np.array_equal(np.isnan(mine), np.isnan(ref))
        assert maxabs(np.nan_to_num(mine), np.nan_to_num(ref)) <= 1e-6


def test_sample_explicit_draws_vs_golden(mods, golden):
    g = golden["igso3_sample"]
    IG = mods["dist"].IsotropicGaussianSO3
    d = IG(torch.tensor(float(g["scalar_eps"]), device=DEV))
    out = d.sample([64], axes=dev(g["scalar_axes"]), unif=dev(g["scalar_unif"]))
    assert out.shape == (64, 3, 3) and maxabs(host(out), g["scalar_out"]) < 1e-5
    d = IG(dev(g["batched_eps"]))
    out = d.sample(axes=dev(g["batched_axes"]), unif=dev(g["batched_unif"]))
    assert maxabs(host(out), g["batched_out"]) < 1e-5          # column-0 quirk reproduced (appendix A.1)
    d2 = IG(dev(g["batched_eps"]), quirk_col0=False)
    out2 = d2.sample(axes=dev(g["batched_axes"]), unif=dev(g["batched_unif"]))
    ref2, _ = O.igso3_sample(g["batched_trap"], g["batched_axes"], g["batched_unif"], row_idx=np.arange(64), weight_row=-1)
    assert maxabs(host(out2), ref2) < 1e-5
    # exact-index parity on the reference's own rows
    smp, ang, _ = mods["B"].igso3_sample(dev(g["batched_trap"]), 64, row_idx=torch.arange(64, device=DEV), quirk_col0=True,
                                         axes=dev(g["batched_axes"]), unif=dev(g["batched_unif"]), want_angle=True)
    _, ang_ref = O.igso3_sample(g["batched_trap"], g["batched_axes"], g["batched_unif"], row_idx=np.arange(64), weight_row=0)
    assert np.array_equal(host(ang), ang_ref)


def test_philox_sampling_is_geometry_independent_and_distributed_right(mods):
    B = mods["B"]
    eps = torch.tensor([0.3], device=DEV)
    trap = B.igso3_build_tables(eps)
    n = 1 << 18
    full, ang, ax = B.igso3_sample(trap, n, seed=7, rng_offset=3, want_angle=True, want_axis=True)
    lo, _, _ = B.igso3_sample(trap, n // 2, seed=7, rng_offset=3, index_base=0)
    hi, _, _ = B.igso3_sample(trap, n - n // 2 - 5, seed=7, rng_offset=3, index_base=n // 2 + 5)
    assert torch.equal(full[: n // 2], lo) and torch.equal(full[n // 2 + 5:], hi)   # shard == slice of the whole
    other, _, _ = B.igso3_sample(trap, n, seed=8, rng_offset=3)
    assert not torch.equal(other, full)
    # angle distribution follows the CDF row: KS distance against the table
    a = np.sort(host(ang))
    k, _ = O.knots()
    cdf = np.interp(a, k[1:], host(trap)[0])
    ks = np.max(np.abs(cdf - (np.arange(n) + 0.5) / n))
    assert ks < 5e-3
    axn = host(ax)
    assert np.abs(np.linalg.norm(axn, axis=1) - 1).max() < 1e-5 and np.abs(axn.mean(0)).max() < 1e-2
    assert abs(float((axn[:, 2] ** 2).mean()) - 1 / 3) < 5e-3


def test_logprob_and_score_vs_golden(mods, golden):
    g = golden["igso3_logprob"]
    R = dev(g["R"])
    for i in range(3):
        d = mods["dist"].IsotropicGaussianSO3(torch.tensor(float(g[f"eps_{i}"]), device=DEV))
        lp = host(d.log_prob(R))
        assert lp.shape == (len(g["R"]), 1)
        ref = g[f"logp_{i}"]
        assert np.max(np.abs(lp - ref) / np.maximum(1, np.abs(ref))) < 1e-5
        _, grad = d.log_prob_and_score(R, dense_grad=True)
        gr = g[f"grad_{i}"]
        scale = np.abs(gr).reshape(len(gr), -1).max(1)[:, None, None] + 1e-3
        assert np.max(np.abs(host(grad) - gr) / scale) < 5e-4
        # ... and the way the reference gets it (distributions.py:189-190): torch.autograd.grad of log_prob wrt the rotations
        Rg = dev(g["R"]).requires_grad_(True)
        (ag,) = torch.autograd.grad(d.log_prob(Rg).sum(), Rg)
        assert np.max(np.abs(host(ag) - gr) / scale) < 5e-4
        wts = torch.linspace(0.5, 2.0, len(gr), device=DEV)[:, None]
        (ag2,) = torch.autograd.grad((d.log_prob(Rg) * wts).sum(), Rg)
        assert torch.allclose(ag2, ag * wts[..., None], rtol=1e-6, atol=0)
        _, sv = d.log_prob_and_score(R)
        ax, ang = O.rmat_to_aa(g["R"], "f64")
        ref_sv = O.igso3_dlogf(ang[:, 0].astype(np.float32), g[f"eps_{i}"])[:, None] * ax
        assert np.max(np.abs(host(sv) - ref_sv) / (np.abs(ref_sv).max(1, keepdims=True) + 1e-3)) < 5e-4
    # per-sample eps (BASELINE config 2b) against the oracle
    rng = np.random.default_rng(1)
    eps = rng.uniform(0.05, 1.0, len(g["R"])).astype(np.float32)
    lp, _, _ = mods["B"].igso3_logprob_score(R, dev(eps))
    lp = host(lp)[:, 0]
    ref = O.igso3_log_prob(g["R"], eps)
    fin = np.isfinite(ref)            # density underflow/overflow -> -inf on both sides (appendix A.4)
    assert np.array_equal(np.isfinite(lp), fin) and np.array_equal(lp[~fin], ref[~fin])
    assert np.max(np.abs(lp[fin] - ref[fin]) / np.maximum(1, np.abs(ref[fin]))) < 2e-5


# ------------------------------------------------------------------ score MLP (G5)
@pytest.mark.parametrize("prec,tol", [("fp32", 2e-5), ("bf16", 2e-2)])
def test_mlp_forward_vs_golden(mods, golden, net, prec, tol):
    g = golden["score_mlp"]
    net.precision = prec
    x, t = dev(g["x"]), dev(g["t"], torch.int64)
    with torch.no_grad():
        out = host(net(x, t))
        out1 = host(net(x, t[:1]))
    net.precision = "fp32"
    scale = np.abs(g["out_64"]).max()
    assert maxabs(out, g["out_64"]) < tol * scale, maxabs(out, g["out_64"])
    assert maxabs(out1, g["out_t1"]) < tol * scale


@pytest.mark.parametrize("prec,tol", [("fp32", 2e-5), ("bf16", 2e-2)])
@pytest.mark.parametrize("n", [1, 31, 32, 33, 1000, 5000])
def test_mlp_forward_ragged_vs_oracle(mods, golden, net, prec, tol, n):
    params = O.flat_params(golden["score_mlp"])
    rng = np.random.default_rng(n)
    R = O.quat_to_rmat(rng.standard_normal((n, 4)).astype(np.float32))
    t = rng.integers(0, 1000, n)
    ref = O.mlp_fwd(params, R, t, "f64")
    net.precision = prec
    with torch.no_grad():
        out = host(net(dev(R), dev(t, torch.int64)))
    net.precision = "fp32"
    assert maxabs(out, ref) < tol * np.abs(ref).max()


@pytest.mark.parametrize("prec,tol", [("fp32", 2e-5), ("bf16", 3e-2)])
def test_mlp_backward_vs_golden(mods, golden, net, prec, tol):
    """MSE-loss gradients through the fused backward (K1 stage, K2 dW GEMM, K3 reduce) vs the reference's autograd."""
    g = golden["score_mlp"]
    net.precision = prec
    x, t = dev(g["x"]), dev(g["t"], torch.int64)
    net.zero_grad()
    out = net(x, t)
    loss = torch.nn.functional.mse_loss(out, dev(g["target"]))
    loss.backward()
    net.precision = "fp32"
    assert abs(float(loss) - float(g["loss"])) < tol * float(g["loss"])
    for l in (0, 2, 4, 6, 8):
        for k in ("weight", "bias"):
            ref = g[f"grad_net_{l}_{k}"]
            mine = host(getattr(net.net[l], k).grad)
            assert mine.shape == ref.shape
            assert maxabs(mine, ref) < tol * np.abs(ref).max() + 1e-9, (l, k, maxabs(mine, ref), np.abs(ref).max())


@pytest.mark.parametrize("n", [1, 33, 1000, 70001])
def test_mlp_backward_ragged_vs_oracle(mods, golden, net, n):
    """ragged sizes, a multi-chunk batch (> 65,536) and the (1,)-shaped t against the fp64 oracle backward"""
    params = O.flat_params(golden["score_mlp"])
    rng = np.random.default_rng(n)
    R = O.quat_to_rmat(rng.standard_normal((n, 4)).astype(np.float32))
    t = rng.integers(0, 1000, n)
    dout = (rng.standard_normal((n, 3)) / n).astype(np.float32)
    B = mods["B"]
    for tt in (t, t[:1]):
        ref = O.mlp_bwd(params, R, tt, dout, "f64")
        mine = host(B.mlp_bwd(dev(params), dev(R), dev(tt, torch.int64), dev(dout), 0))
        assert maxabs(mine, ref) < 3e-5 * np.abs(ref).max(), (n, maxabs(mine, ref), np.abs(ref).max())


@pytest.mark.parametrize("prec,tol", [("fp32", 2e-5), ("bf16", 3e-2)])
def test_mlp_table_path_matches_in_kernel_embedding(mods, golden, net, prec, tol):
    """t_table > 0 (per-timestep effective-bias / embedding tables gathered by t) must agree with the
    in-kernel embedding path and with the oracle, forward and backward."""
    B = mods["B"]
    code = 0 if prec == "fp32" else 1
    params_np = O.flat_params(golden["score_mlp"])
    params = dev(params_np)
    rng = np.random.default_rng(12)
    n = 3000
    R = O.quat_to_rmat(rng.standard_normal((n, 4)).astype(np.float32))
    t = rng.integers(0, 1000, n)
    dout = (rng.standard_normal((n, 3)) / n).astype(np.float32)
    ref = O.mlp_fwd(params_np, R, t, "f64")
    for tab in (0, 1000, 1500):
        out = host(B.mlp_fwd(params, dev(R), dev(t, torch.int64), code, t_table=tab))
        assert maxabs(out, ref) < tol * np.abs(ref).max(), (tab, maxabs(out, ref))
    gref = O.mlp_bwd(params_np, R, t, dout, "f64")
    for tab in (0, 1000):
        g = host(B.mlp_bwd(params, dev(R), dev(t, torch.int64), dev(dout), code, t_table=tab))
        assert maxabs(g, gref) < max(tol, 3e-5) * np.abs(gref).max(), (tab, maxabs(g, gref))
    # shared timestep ((1,)-shaped t) through the table path
    out1 = host(B.mlp_fwd(params, dev(R), dev(t[:1], torch.int64), code, t_table=1000))
    assert maxabs(out1, O.mlp_fwd(params_np, R, t[:1], "f64")) < tol * np.abs(ref).max()


def test_mlp_backward_is_additive_across_stash_chunks(mods, golden, net):
    """n > 2^19 crosses the backward's stash-chunk boundary: the gradient of the whole batch must equal the sum
    of the gradients of two halves (linearity in dout) -- a size-independent property at BASELINE scale."""
    B = mods["B"]
    n = (1 << 19) + 999
    g = torch.Generator(device=DEV).manual_seed(4)
    R = B.quat_to_rmat(torch.randn(n, 4, device=DEV, generator=g))
    t = torch.randint(0, 1000, (n,), device=DEV, generator=g)
    dout = torch.randn(n, 3, device=DEV, generator=g) / n
    params = net.flat_params_nograd()
    for prec, tol in ((0, 2e-5), (1, 2e-3)):
        full = B.mlp_bwd(params, R, t, dout, prec)
        cut = 300000
        parts = B.mlp_bwd(params, R[:cut], t[:cut], dout[:cut], prec) + B.mlp_bwd(params, R[cut:], t[cut:], dout[cut:], prec)
        assert float((full - parts).abs().max()) < tol * float(full.abs().max())
        assert torch.isfinite(full).all()


def test_training_step_matches_reference_gradients(mods, golden, net):
    """One full SO3Diffusion training step (p_losses -> backward) with the reference's recorded draws."""
    g = golden["train_step"]
    for T in (100, 1000):
        pre = f"T{T}_s0_"
        proc = mods["diff"].SO3Diffusion(net, timesteps=T, betas=golden["schedule"][f"betas64_{T}"]).to(DEV)
        net.zero_grad()
        loss = proc.p_losses(dev(g[pre + "x0"]), dev(g[pre + "t"], torch.int64), axes=dev(g[pre + "axes"]),
                             unif=dev(g[pre + "unif"]))
        loss.backward()
        flat = torch.cat([p.grad.reshape(-1) for p in net.parameters()])
        ref = g[pre + "grad_flat"]
        assert abs(float(loss) - float(g[pre + "loss"])) < 2e-5 * float(g[pre + "loss"])
        assert maxabs(host(flat), ref) < 1e-4 * np.abs(ref).max()
    # and an optimizer step moves the loss down on a fixed batch (plumbing check of the autograd bridge)
    torch.manual_seed(0)
    import copy
    net2 = copy.deepcopy(net)
    proc = mods["diff"].SO3Diffusion(net2, timesteps=100).to(DEV)
    opt = torch.optim.Adam(net2.parameters(), lr=3e-3)
    x0, t = dev(g["T100_s0_x0"]), dev(g["T100_s0_t"], torch.int64)
    ax, un = dev(g["T100_s0_axes"]), dev(g["T100_s0_unif"])
    first = last = None
    for _ in range(30):
        loss = proc.p_losses(x0, t, axes=ax, unif=un)
        opt.zero_grad()
        loss.backward()
        opt.step()
        first = float(loss) if first is None else first
        last = float(loss)
    assert last < 0.7 * first


# ------------------------------------------------------------------ A12 forward noising + target
@pytest.mark.parametrize("T", [100, 1000])
@pytest.mark.parametrize("seed", [0, 1, 2])
def test_p_losses_pieces_vs_golden(mods, golden, net, T, seed):
    g = golden["train_step"]
    pre = f"T{T}_s{seed}_"
    proc = mods["diff"].SO3Diffusion(net, timesteps=T, betas=golden["schedule"][f"betas64_{T}"]).to(DEV)
    trap_q, _ = proc._tables()
    x0, t = dev(g[pre + "x0"]), dev(g[pre + "t"], torch.int64)
    x_t, target, noise = mods["B"].q_sample_target(proc._sched, trap_q, x0, t, quirk_col0=True, axes=dev(g[pre + "axes"]),
                                                   unif=dev(g[pre + "unif"]), want_noise=True)
    # natively built rows can flip an index for ~0.02% of draws (8c); the fixtures here do not hit one
    assert maxabs(host(noise), g[pre + "noise"]) < 2e-5
    assert maxabs(host(x_t), g[pre + "x_t"]) < 2e-5
    assert maxabs(host(target), g[pre + "target"]) < 1e-5 * max(1.0, np.abs(g[pre + "target"]).max())
    # teacher-forced noise path == reference q_sample(x_start, t, noise)
    x_t2 = proc.q_sample(x0, t, noise=dev(g[pre + "noise"]))
    assert maxabs(host(x_t2), g[pre + "x_t"]) < 2e-5
    with torch.no_grad():
        loss = proc.p_losses(x0, t, axes=dev(g[pre + "axes"]), unif=dev(g[pre + "unif"]))
    assert abs(float(loss) - float(g[pre + "loss"])) < 2e-5 * float(g[pre + "loss"])


# ------------------------------------------------------------------ A13 reverse steps (G2)
@pytest.mark.parametrize("tval", [0, 1, 50, 500, 950, 998, 999])
def test_p_sample_step_vs_golden_G2(mods, golden, net, tval):
    g = golden["p_sample_steps"]
    pre = f"t{tval}_"
    proc = mods["diff"].SO3Diffusion(net, timesteps=1000, betas=golden["schedule"]["betas64_1000"]).to(DEV)
    x = dev(g["x"])
    n = x.shape[0]
    # rotation math with the reference's fp32 network output teacher-forced
    x0h, mean = mods["B"].p_mean(proc._sched, x, dev(g[pre + "v"]), tval, want_x0hat=True)
    err = frob_err(host(mean), g[pre + "mean_64f"])
    ref_err = frob_err(g[pre + "mean"], g[pre + "mean_64f"])
    # G2, PER SAMPLE: err_i <= max(1e-5, 2 ref_err_i), the reference's own fp32-vs-fp64 error on that sample.  Both are
    # rounding noise amplified by the same conditioning (up to 2e4 at t = 999), but they are INDEPENDENT draws of it, so a sample
    # on which the reference happened to round luckily may exceed its own limit: the survey's outlier budget bounds how many
    # (<= 2 % for t <= 998, 40 % at t = 999, measured on the reference's own fp32 path), and NO sample may exceed what its
    # conditioning allows (conftest.reverse_step_bound with an exact network: the derived per-sample tolerance of the chain
    # tests) -- one badly conditioned reference sample no longer licenses every device sample.
    from conftest import reverse_step_bound
    lim = np.maximum(1e-5, 2 * ref_err)
    over = err > lim
    # (the survey's 2 % / 40 % are RATES measured on many samples; this fixture has n = 64, where one binomial standard error is
    #  1.75 % / 6.1 %: the budgets are the survey's rates plus at most one standard error -- 3 % = one sample, 45 % = 28 samples)
    p_out = 0.40 if tval == 999 else 0.02
    assert (0.45 if tval == 999 else 0.03) <= p_out + np.sqrt(p_out * (1 - p_out) / n) + 1e-9
    assert over.mean() <= (0.45 if tval == 999 else 0.03), (tval, over.mean())
    sch = host(proc._sched)
    coef = tuple(float(sch[i][tval]) for i in (6, 7, 10, 11))
    om_x = O.rmat_to_aa(g["x"], "f64")[1][:, 0]
    om_h = O.rmat_to_aa(g[pre + "x0hat_64f"], "f64")[1][:, 0]
    bound = reverse_step_bound(coef, om_x, om_h, dv=0.0)
    assert (err <= np.maximum(lim, bound)).all(), (tval, float((err / np.maximum(lim, bound)).max()))
    assert np.median(err) <= max(2e-6, 2 * np.median(ref_err))
    # fused step (MLP + mean + noise) with explicit draws, fp32 network
    t = torch.full((n,), tval, device=DEV, dtype=torch.long)
    kw = dict(axes=dev(g[pre + "axes"]), unif=dev(g[pre + "unif"])) if tval > 0 else {}
    out = host(proc.p_sample(x, t, **kw))
    err = frob_err(out, g[pre + "xprev_64f"])
    ref_err = frob_err(g[pre + "xprev"], g[pre + "xprev_64f"])
    assert err.max() <= max(2e-5, 4 * ref_err.max() + 1e-3 * (tval >= 998)), (err.max(), ref_err.max())
    # (1,)-shaped t (so3_test.py:31) gives the same result
    out1 = host(proc.p_sample(x, t[:1], **kw))
    assert np.array_equal(out, out1)


@pytest.mark.parametrize("tval", [0, 1, 50, 500, 950, 998, 999])
def test_p_sample_step_outlier_rates_G2_n1024(mods, golden, net, tval):
    """G2's outlier budgets as the survey states them -- at most 2 % of the samples (t <= 998) / 40 % (t = 999) further from the
    float64 reference than max(1e-5, 2 x the reference's own fp32 error on that sample) -- on a fixture large enough
    for the rates to apply WITHOUT a small-sample margin (n = 1024: tools/make_golden.py p_sample_steps_large; the n = 64 fixture
    above carries one binomial standard error).  No sample beyond its conditioning-derived bound either."""
    from conftest import reverse_step_bound
    g = golden["p_sample_steps_n1024"]
    pre = f"t{tval}_"
    proc = mods["diff"].SO3Diffusion(net, timesteps=1000, betas=golden["schedule"]["betas64_1000"]).to(DEV)
    x = dev(g["x"])
    assert x.shape[0] == 1024
    _, mean = mods["B"].p_mean(proc._sched, x, dev(g[pre + "v"]), tval, want_x0hat=True)
    err = frob_err(host(mean), g[pre + "mean_64f"])
    ref_err = frob_err(g[pre + "mean"], g[pre + "mean_64f"])
    lim = np.maximum(1e-5, 2 * ref_err)
    over = err > lim
    assert over.mean() <= (0.40 if tval == 999 else 0.02), (tval, float(over.mean()))
    sch = host(proc._sched)
    coef = tuple(float(sch[i][tval]) for i in (6, 7, 10, 11))
    om_x = O.rmat_to_aa(g["x"], "f64")[1][:, 0]
    om_h = O.rmat_to_aa(g[pre + "x0hat_64f"], "f64")[1][:, 0]
    bound = reverse_step_bound(coef, om_x, om_h, dv=0.0)
    assert (err <= np.maximum(lim, bound)).all(), (tval, float((err / np.maximum(lim, bound)).max()))
    assert np.median(err) <= max(2e-6, 2 * np.median(ref_err))


def test_chain_explicit_draws_vs_golden(mods, golden, net):
    g = golden["p_sample_chain"]
    T = len(g["betas"])
    proc = mods["diff"].SO3Diffusion(net, betas=g["betas"]).to(DEV)
    d = mods["dist"].IsotropicGaussianSO3(torch.ones([], device=DEV))
    x = d.sample((16,), axes=dev(g["axes"][0]), unif=dev(g["unif"][0]))
    for step, t in enumerate(reversed(range(T))):
        kw = dict(axes=dev(g["axes"][step + 1]), unif=dev(g["unif"][step + 1])) if t > 0 else {}
        x = proc.p_sample(x, torch.full((16,), t, device=DEV, dtype=torch.long), **kw)
    assert maxabs(host(x), g["x_final"]) < 5e-4
    assert maxabs(host(x @ x.transpose(-1, -2)), np.eye(3)[None]) < 1e-4


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_chain_kernel_equals_stepwise_and_shards(mods, golden, net, prec):
    """One launch over a span == single-step launches (same Philox counters) to rounding, runs are bit-reproducible,
    and a 2-way sharded run (index_base) == the unsharded run bit-for-bit: results do not depend on GPU count."""
    from so3x import rng
    net.precision = prec
    T = 50
    proc = mods["diff"].SO3Diffusion(net, timesteps=T).to(DEV)
    n = 1000
    gen = torch.Generator(device=DEV).manual_seed(3)
    x0 = mods["util"].quat_to_rmat(torch.randn(n, 4, device=DEV, generator=gen))
    rng.manual_seed(11)
    full = proc.p_sample_loop((n,), x_init=x0)
    _, trap_p = proc._tables()
    B = mods["B"]
    params = net.flat_params_nograd()
    # one launch over a span of the chain == single-step launches over the same span, up to the matrix <-> unit
    # quaternion conversion at every launch boundary (the state is a quaternion inside a launch).  Checked below
    # t = 30, where no step amplifies a 1e-7 perturbation by more than ~2.
    span = B.p_sample_chain(params, proc._sched, trap_p, x0, 30, 31, seed=11, rng_offset=0, precision=net.precision_code)
    x = x0
    for t in reversed(range(31)):
        x = B.p_sample_chain(params, proc._sched, trap_p, x, t, 1, seed=11, rng_offset=0, precision=net.precision_code)
    assert float((x - span).abs().max()) < (1e-4 if prec == "fp32" else 2e-2)
    again = proc_again = B.p_sample_chain(params, proc._sched, trap_p, x0, T - 1, T, seed=11, rng_offset=0,
                                          precision=net.precision_code)
    assert torch.equal(again, full)   # bit-reproducible
    a = B.p_sample_chain(params, proc._sched, trap_p, x0[:600], T - 1, T, seed=11, rng_offset=0, index_base=0,
                         precision=net.precision_code)
    b = B.p_sample_chain(params, proc._sched, trap_p, x0[600:], T - 1, T, seed=11, rng_offset=0, index_base=600,
                         precision=net.precision_code)
    assert torch.equal(torch.cat([a, b]), full)
    assert float((full @ full.transpose(-1, -2) - torch.eye(3, device=DEV)).abs().max()) < 1e-4
    assert not torch.isnan(full).any()
    net.precision = "fp32"


def test_chain_fp32_vs_oracle_stepwise(mods, golden, net):
    """Device chain with in-kernel Philox noise vs the CPU oracle fed the SAME noise (recovered through the
    sampler's angle/axis outputs, same Philox counters).  Every step is checked from the device state, so
    the ill-conditioned steps (t ~ T-1, scale up to 2e4) do not compound into the well-conditioned ones."""
    B = mods["B"]
    T = 30
    betas = O.cosine_beta_schedule(T)
    proc = mods["diff"].SO3Diffusion(net, betas=betas).to(DEV)
    _, trap_p = proc._tables()
    sched = O.schedule_from_betas(betas)
    params_np = O.flat_params(golden["score_mlp"])
    n = 256
    x0 = O.quat_to_rmat(np.random.default_rng(5).standard_normal((n, 4)).astype(np.float32))
    xd = dev(x0)
    for t in reversed(range(T)):
        xh = host(xd)
        xd = B.p_sample_chain(net.flat_params_nograd(), proc._sched, trap_p, xd, t, 1, seed=5, rng_offset=100, precision=0)
        coef = [float(sched[i][t]) for i in (6, 7, 10, 11)]
        v = O.mlp_fwd(params_np, xh, np.full(n, t), "f64")
        x0h, ref = O.p_mean(xh, v, *coef, "f64")
        if t > 0:
            _, ang, ax = B.igso3_sample(trap_p, n, row_const=t, seed=5, rng_offset=100 + t, want_angle=True, want_axis=True)
            ref = O.rmul(ref, O.aa_to_rmat(host(ax), host(ang), "f64"), "f64")
        # conditioning of the two logs: error ~ 1e-7 * scale / (pi - omega)
        _, a1 = O.rmat_to_aa(xh, "f64")
        _, a2 = O.rmat_to_aa(x0h, "f64")
        cond = max(coef[0], 1.0) * (1.0 / (np.pi - a1[:, 0]) + 1.0 / (np.pi - a2[:, 0]) + 1.0)
        err = np.abs(host(xd) - ref).reshape(n, -1).max(1)
        assert (err <= 2e-5 + 4e-6 * cond).all(), (t, float(err.max()), float(cond[np.argmax(err)]))
    assert maxabs(host(xd @ xd.transpose(-1, -2)), np.eye(3)[None]) < 1e-4


@pytest.mark.parametrize("prec", [1, 2], ids=["bf16", "f16-leg"])
@pytest.mark.parametrize("t", [5, 300, 700])
def test_chain_bf16_step_vs_oracle(mods, golden, net, t, prec):
    """bf16-operand chain step vs the fp64 oracle with the same Philox noise.  Covers both MFMA tiles of a wave
    (lanes 0..31 / 32..63 take different operand-exchange paths) and ragged tails."""
    B = mods["B"]
    T = 1000
    proc = mods["diff"].SO3Diffusion(net, timesteps=T).to(DEV)
    _, trap_p = proc._tables()
    sched = O.schedule_from_betas(O.cosine_beta_schedule(T))
    params_np = O.flat_params(golden["score_mlp"])
    n = 200
    x0 = O.quat_to_rmat(np.random.default_rng(t).standard_normal((n, 4)).astype(np.float32))
    out = host(B.p_sample_chain(net.flat_params_nograd(), proc._sched, trap_p, dev(x0), t, 1, seed=9, rng_offset=7, precision=prec))
    coef = [float(sched[i][t]) for i in (6, 7, 10, 11)]
    v = O.mlp_fwd(params_np, x0, np.full(n, t), "f64")
    _, ref = O.p_mean(x0, v, *coef, "f64")
    _, ang, ax = B.igso3_sample(trap_p, n, row_const=t, seed=9, rng_offset=7 + t, want_angle=True, want_axis=True)
    ref = O.rmul(ref, O.aa_to_rmat(host(ax), host(ang), "f64"), "f64")
    err = np.abs(out - ref).reshape(n, -1).max(1)
    # bf16 operands: ~2e-2 relative on v (|v| ~ 0.3 with the seed-0 weights) times b_t <= 7 at these t
    assert err.max() < 2e-2 * max(1.0, coef[1]), (t, err.max())
    assert np.median(err) < 4e-3 * max(1.0, coef[1])
    # the two halves of a wave must agree in quality (a broken operand exchange shows up as one bad half)
    lanes = np.arange(n) % 64
    assert abs(np.median(err[lanes < 32]) - np.median(err[lanes >= 32])) < 2e-3 * max(1.0, coef[1])


@pytest.mark.parametrize("prec", [1, 2], ids=["bf16", "f16-leg"])
@pytest.mark.parametrize("t", [950, 998, 999])
def test_chain_bf16_step_vs_oracle_at_the_head_of_the_chain(mods, golden, net, t, prec):
    """the SHIPPED bf16 chain kernel (256-entry SiLU table, hardware v_sin / v_cos behind v_fract range
    reduction, layer-0 bias split into bf16 + remainder) step-wise against the f64 oracle with the same Philox noise where the
    reverse step is hardest: t >= 950, where x0hat = so3_scale(x_t, a_t) @ exp(-b_t v) scales a matrix log by up to 20291 and
    the Rodrigues angle reaches 1e4 revolutions (reference diffusion.py:291-326, util.py:349-361).  The tolerance is derived in
    conftest.reverse_step_bound: fp32 conditioning of the two logs + the bf16 network's error dv carried through
    exp(b_t .), log and the c1_t scaling, saturating at c1_t * 2 pi.  dv = 2e-2 absolute (bf16 operands through five layers:
    ~2e-2 relative on |v| ~ 0.3, test_chain_bf16_step_vs_oracle; generous on purpose -- at t >= 998 the bound has saturated
    anyway, at t = 950 it contributes 1.4e-3).  A wrong schedule coefficient, noise row, trig range reduction or operand
    exchange shows up as O(0.1 - 1) errors, far outside it."""
    from conftest import reverse_step_bound
    B = mods["B"]
    T = 1000
    proc = mods["diff"].SO3Diffusion(net, timesteps=T).to(DEV)
    _, trap_p = proc._tables()
    sched = O.schedule_from_betas(O.cosine_beta_schedule(T))
    params_np = O.flat_params(golden["score_mlp"])
    n = 640 + 37   # ten full waves and a ragged one
    x0 = O.quat_to_rmat(np.random.default_rng(t).standard_normal((n, 4)).astype(np.float32))
    out = host(B.p_sample_chain(net.flat_params_nograd(), proc._sched, trap_p, dev(x0), t, 1, seed=9, rng_offset=7, precision=prec))
    coef = [float(sched[i][t]) for i in (6, 7, 10, 11)]
    v = O.mlp_fwd(params_np, x0, np.full(n, t), "f64")
    x0h, ref = O.p_mean(x0, v, *coef, "f64")
    _, ang, ax = B.igso3_sample(trap_p, n, row_const=t, seed=9, rng_offset=7 + t, want_angle=True, want_axis=True)
    ref = O.rmul(ref, O.aa_to_rmat(host(ax), host(ang), "f64"), "f64")
    _, a1 = O.rmat_to_aa(x0, "f64")
    _, a2 = O.rmat_to_aa(x0h, "f64")
    bound = reverse_step_bound(coef, a1[:, 0], a2[:, 0], dv=2e-2)
    err = np.abs(out - ref).reshape(n, -1).max(1)
    assert np.isfinite(out).all()
    assert (err <= bound).all(), (t, float(err.max()), float(bound[np.argmax(err - bound)]))
    # the gate is never vacuous: c1_t * 2 pi <= 1.95e-2 plus the fp32 terms (a sample whose x_t sits within 1e-4 of angle pi has an
    # ill-conditioned log in ANY fp32 arithmetic, the reference's included: allowed for 1 % of the batch)
    assert np.quantile(bound, 0.99) < 2.5e-2 and np.median(bound) < (2.5e-2 if t >= 998 else 5e-3)
    assert np.abs(out @ out.transpose(0, 2, 1) - np.eye(3)).max() < 1e-4
    # the two halves of a wave must agree in quality (a broken operand exchange shows up as one bad half)
    lanes = np.arange(n) % 64
    assert abs(np.median(err[lanes < 32]) - np.median(err[lanes >= 32])) < 0.5 * np.median(err) + 1e-5
    # and the fp32 parity kernel on the same step sits inside the dv = 0 bound (same derivation, no network term)
    out32 = host(B.p_sample_chain(net.flat_params_nograd(), proc._sched, trap_p, dev(x0), t, 1, seed=9, rng_offset=7, precision=0))
    err32 = np.abs(out32 - ref).reshape(n, -1).max(1)
    b32 = reverse_step_bound(coef, a1[:, 0], a2[:, 0], dv=2e-6)
    assert (err32 <= b32).all(), (t, float(err32.max()), float(b32[np.argmax(err32 - b32)]))


def _ab_chain(proc, trap_p, params, x, t_start, n_steps, seed, rng_offset):
    """so3x_p_sample_chain (bf16) of the A/B build libso3x_ab.so through its raw C ABI.  The A/B build is the same ABI compiled
    with -DSO3X_AB_BUILD: the only library whose launcher reads the SO3X_AB_* environment switches (csrc/so3x_diffusion.hip);
    the library the package loads has ONE bf16 form and no getenv."""
    import ctypes as C
    from so3x import backend as B
    path = os.path.join(os.path.dirname(B.LIB_PATH), "libso3x_ab.so")
    if not hasattr(_ab_chain, "lib"):
        _ab_chain.lib = C.CDLL(path)
    lib = _ab_chain.lib
    lib.so3x_p_sample_workspace_bytes.restype = C.c_size_t
    T = proc._sched.shape[1]
    nb = lib.so3x_p_sample_workspace_bytes(C.c_int(T), C.c_int(1))
    ws = torch.empty(nb, dtype=torch.uint8, device=DEV)
    out = torch.empty_like(x)
    P = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    rc = lib.so3x_p_sample_chain(C.c_void_p(torch.cuda.current_stream().cuda_stream), P(params), P(proc._sched), C.c_int(T), P(trap_p),
                                 P(proc._guide_p), P(x), P(out), C.c_int(t_start), C.c_int(n_steps), None, None, C.c_uint64(seed),
                                 C.c_uint64(rng_offset), C.c_int64(0), C.c_int64(x.shape[0]), C.c_int(1), P(ws), C.c_size_t(nb))
    assert rc == 0, rc
    torch.cuda.synchronize()
    return out


_AB_KEYS = ("SO3X_AB_TAB", "SO3X_AB_PAIR", "SO3X_AB_TRIG", "SO3X_AB_BLOCK", "SO3X_AB_CDF")


def test_chain_bf16_kernel_forms_agree(mods, net, monkeypatch):
    """The shipped bf16 chain kernel (lane-replicated SiLU table whose address is the convert's result, stages laid out MFMA
    gap by MFMA gap) against its other forms, which exist only in the A/B build (libso3x_ab.so, loaded here by path): the 2 KB
    table with shift-add addressing must give the SAME BITS (same index, same entry, same multiply-add -- only the addressing
    differs); one tile after the other instead of the paired stream likewise (same arithmetic per tile), and so must the reverse
    step's inverse-CDF search on global memory instead of the per-wave LDS record the kernel stages by LDS-DMA; Cody-Waite
    instead of hardware sine / cosine agrees to the trigonometry's 1e-6 per step.  The product library (no switches) must give
    the bits of the A/B build's default form AND must not react to the switches.  Ragged sizes cover partial waves and a
    partial last workgroup."""
    B = mods["B"]
    proc = mods["diff"].SO3Diffusion(net, timesteps=1000).to(DEV)
    _, trap_p = proc._tables()
    params = net.flat_params_nograd()
    for n in (1, 63, 65, 4100, 70000):
        x = mods["util"].quat_to_rmat(torch.randn(n, 4, device=DEV, generator=torch.Generator(device=DEV).manual_seed(n)))
        run = lambda: _ab_chain(proc, trap_p, params, x, 420, 12, 3, 5)  # noqa: E731
        for k in _AB_KEYS:
            monkeypatch.delenv(k, raising=False)
        base = run()
        assert torch.isfinite(base).all()
        shipped = B.p_sample_chain(params, proc._sched, trap_p, x, 420, 12, seed=3, rng_offset=5, precision=1)
        assert torch.equal(shipped, base), n
        monkeypatch.setenv("SO3X_AB_TAB", "narrow")
        assert torch.equal(run(), base), n
        monkeypatch.delenv("SO3X_AB_TAB")
        monkeypatch.setenv("SO3X_AB_PAIR", "0")
        assert torch.equal(run(), base), n
        monkeypatch.delenv("SO3X_AB_PAIR")
        monkeypatch.setenv("SO3X_AB_BLOCK", "256")
        assert torch.equal(run(), base), n
        monkeypatch.delenv("SO3X_AB_BLOCK")
        monkeypatch.setenv("SO3X_AB_CDF", "global")   # inverse-CDF search on global memory instead of the LDS-staged record
        assert torch.equal(run(), base), n
        monkeypatch.delenv("SO3X_AB_CDF")
        monkeypatch.setenv("SO3X_AB_TRIG", "cw")
        d = (run() - base).abs().reshape(n, -1).max(1).values
        assert float(d.median()) < 2e-5 and float(d.quantile(0.99) if n > 100 else d.max()) < 2e-3, (n, float(d.max()))
        # the product library has no such switch: same bits with the variable exported
        assert torch.equal(B.p_sample_chain(params, proc._sched, trap_p, x, 420, 12, seed=3, rng_offset=5, precision=1), base), n
        monkeypatch.delenv("SO3X_AB_TRIG")


@pytest.mark.parametrize("t_start", [999, 960])
def test_chain_bf16_hardware_trig_at_the_head_of_the_chain(mods, net, monkeypatch, t_start):
    """the hardware sine / cosine (v_sin / v_cos behind v_fract range reduction, csrc/so3x_math.hpp) against
    Cody-Waite where the Rodrigues angle is LARGEST: at t >= 950 the reverse step scales a matrix log by
    sqrt_recip_alphas_cumprod up to 20291 (reference diffusion.py:291-297, util.py:349-361), i.e. arguments of up to ~1e4
    revolutions go through the range reduction.  Ten steps from t = 999 (and from 960) with the two trigonometries on the same
    state and noise, step by step so that the comparison is of ONE step (the map itself amplifies differences of the state by
    up to 2e4 there, which is the reference's conditioning, not the trigonometry's).

    Tolerance: v_fract(theta / 2 pi) keeps the fraction of a 24-bit float: at theta = 6.4e4 rad (1e4 revolutions) one ulp of
    the argument is 2^-10 of a revolution... for BOTH forms -- the product theta = scale * |log| is rounded to fp32 before
    either reduction sees it, so the two can differ only by their own reduction error: Cody-Waite ~1e-7 absolute, v_fract +
    v_sin ~2^-23 of a revolution x 2 pi = 7.5e-7 of angle plus the 1e-6 of v_sin itself.  Gate: median 5e-6, 99th percentile
    2e-4 (a sample near the log's pi branch amplifies through 1 / (pi - omega)), per step."""
    proc = mods["diff"].SO3Diffusion(net, timesteps=1000).to(DEV)
    _, trap_p = proc._tables()
    params = net.flat_params_nograd()
    n = 4096
    x = mods["util"].quat_to_rmat(torch.randn(n, 4, device=DEV, generator=torch.Generator(device=DEV).manual_seed(77)))
    for k in _AB_KEYS:
        monkeypatch.delenv(k, raising=False)
    for s in range(10):
        t = t_start - s
        hw = _ab_chain(proc, trap_p, params, x, t, 1, 3, 5)
        monkeypatch.setenv("SO3X_AB_TRIG", "cw")
        cw = _ab_chain(proc, trap_p, params, x, t, 1, 3, 5)
        monkeypatch.delenv("SO3X_AB_TRIG")
        assert torch.isfinite(hw).all() and torch.isfinite(cw).all()
        d = (hw - cw).abs().reshape(n, -1).max(1).values
        assert float(d.median()) < 5e-6 and float(d.quantile(0.99)) < 2e-4, (t, float(d.median()), float(d.quantile(0.99)), float(d.max()))
        x = hw


@pytest.mark.parametrize("prec", ["fp32", "bf16", "bf16+f16-operand chain"])
@pytest.mark.parametrize("fixture", ["chain_samples_trained", "chain_samples_T1000"])
def test_G3_full_chain_two_sample_test_vs_reference(mods, golden, net, prec, fixture):
    """Gate G3: 4096 samples of the full 1000-step chain must be statistically
    indistinguishable from 4096 samples of the REFERENCE's own p_sample_loop under the reference's kernel
    two-sample test (util.MMD + rmat_gaussian_kernel + Ker_2samp_test, alpha = 0.05), stay orthonormal and finite.
    Fixtures (tools/make_golden.py): `chain_samples_trained` = a RotPredict trained for 3000 steps BY THE
    REFERENCE on its two-mode data (so3_train.py:65-72) -- a concentrated population, the test has power;
    `chain_samples_T1000` = the seed-0 untrained net (near-uniform population)."""
    import copy
    from so3x import rng
    g = golden[fixture]
    ref = g["x_final"]
    m = len(ref)
    mynet = copy.deepcopy(net)
    if "net_0_weight" in g:
        mynet.load_state_dict({f"net.{l}.{k}": torch.from_numpy(g[f"net_{l}_{k}"]) for l in (0, 2, 4, 6, 8) for k in ("weight", "bias")})
    mynet = mynet.to(DEV)
    mynet.precision = prec.split("+")[0]
    mynet.chain_operands = "f16" if "f16" in prec else None   # round 4's extra leg: the bf16 chain with IEEE half operand bits
    proc = mods["diff"].SO3Diffusion(mynet, timesteps=1000).to(DEV)
    rng.manual_seed(2024)
    x = proc.p_sample_loop((m,))
    assert not torch.isnan(x).any()
    assert float((x @ x.transpose(-1, -2) - torch.eye(3, device=DEV)).abs().max()) < 1e-5
    mine = host(x)
    thr = O.ker_2samp_threshold(m)
    mmd = O.MMD(mine, ref)
    assert mmd < thr, (mmd, thr)
    assert mmd < 5 * max(O.MMD(ref[: m // 2], ref[m // 2:]), 1e-3)  # tighter than the reference's bound: ~ estimator noise
    if fixture == "chain_samples_trained":
        # power: Haar-uniform rotations are rejected against this population
        uni = O.quat_to_rmat(np.random.default_rng(0).standard_normal((m, 4)), "f64")
        assert O.MMD(uni, ref) > thr
        # same concentration around the two training modes (so3_train.py:65-72) and the same mode split
        z90 = np.array([[0.0, -1.0, 0.0], [1.0, 0.0, 0.0], [0.0, 0.0, 1.0]])
        def stats(X):
            d0 = O.rmat_dist(X, np.broadcast_to(z90, X.shape).copy(), "f64")
            d1 = O.rmat_dist(X, np.broadcast_to(z90.T, X.shape).copy(), "f64")
            return np.median(np.minimum(d0, d1)), np.mean(d0 < d1)
        med_r, frac_r = stats(ref.astype(np.float64))
        med_m, frac_m = stats(mine.astype(np.float64))
        assert abs(med_m - med_r) < 0.25 * med_r, (med_m, med_r)
        assert abs(frac_m - frac_r) < 4 * np.sqrt(0.25 / m) + 0.01, (frac_m, frac_r)


@pytest.mark.parametrize("prec,graph", [("fp32", False), ("bf16", False), ("bf16", True)])
def test_training_with_this_stack_reaches_the_reference_trained_population(mods, golden, prec, graph):
    """End-to-end training parity: the recipe that made `chain_samples_trained` (tools/make_golden.py: the REFERENCE trained
    with its own loss for 3000 steps of batch 256, Adam 1e-3, on the two-mode data of so3_train.py:65-72, then sampled
    with its own p_sample_loop) is run here with THIS stack's noising, loss, backward and sampler -- a different
    initialisation and different noise, so the learned populations can only agree statistically: the reference's kernel
    two-sample test must not tell them apart, and the concentration around the modes and the mode split must match."""
    from so3x import rng
    from so3x.so3_train import RotPredict
    from so3x.graphs import TrainStepGraph
    ref = golden["chain_samples_trained"]["x_final"]
    m = len(ref)
    torch.manual_seed(0)
    rng.manual_seed(77)
    mynet = RotPredict(out_type="skewvec", precision=prec).to(DEV)
    proc = mods["diff"].SO3Diffusion(mynet, timesteps=1000).to(DEV)
    opt = torch.optim.Adam(mynet.parameters(), lr=1e-3, fused=True, capturable=graph)
    z90 = torch.tensor([[0.0, -1.0, 0.0], [1.0, 0.0, 0.0], [0.0, 0.0, 1.0]], device=DEV)
    rotations = torch.stack((z90, z90.T), dim=0)
    gen = torch.Generator(device=DEV).manual_seed(5)
    stepper = TrainStepGraph(proc, opt, (256, 3, 3)) if graph else None
    for i in range(3000):
        x0 = rotations[torch.randint(0, 2, (256,), device=DEV, generator=gen)]
        if graph:
            loss = stepper.step(x0)
        else:
            loss = proc(x0)
            opt.zero_grad()
            loss.backward()
            opt.step()
    assert float(loss.detach()) < 1.0
    proc.rng_counter = None
    x = proc.p_sample_loop((m,))
    assert not torch.isnan(x).any()
    mine = host(x)
    thr = O.ker_2samp_threshold(m)
    mmd = O.MMD(mine, ref)
    assert mmd < thr, (mmd, thr)
    z = np.array([[0.0, -1.0, 0.0], [1.0, 0.0, 0.0], [0.0, 0.0, 1.0]])

    def stats(X):
        d0 = O.rmat_dist(X, np.broadcast_to(z, X.shape).copy(), "f64")
        d1 = O.rmat_dist(X, np.broadcast_to(z.T, X.shape).copy(), "f64")
        return np.median(np.minimum(d0, d1)), np.mean(d0 < d1)
    med_r, frac_r = stats(ref.astype(np.float64))
    med_m, frac_m = stats(mine.astype(np.float64))
    print(f"[{prec}{' graph' if graph else ''}] MMD {mmd:.2e} (bound {thr:.2e}); median mode distance {med_m:.4f} vs reference {med_r:.4f}; split {frac_m:.3f} vs {frac_r:.3f}")
    assert med_m < 2.0 * med_r and med_m > 0.4 * med_r, (med_m, med_r)   # two short trainings: same scale of spread
    assert abs(frac_m - 0.5) < 0.1 and abs(frac_r - 0.5) < 0.1, (frac_m, frac_r)
    uni = O.quat_to_rmat(np.random.default_rng(0).standard_normal((m, 4)), "f64")
    assert O.MMD(uni, mine) > thr                                             # and it is nothing like Haar-uniform


def test_full_size_chain_properties(mods, net):
    """BASELINE config 3 shape (2^20 rotations), a short bf16 chain: no NaN, orthonormal, det +1."""
    net.precision = "bf16"
    proc = mods["diff"].SO3Diffusion(net, timesteps=1000).to(DEV)
    n = 1 << 20
    x0 = mods["util"].quat_to_rmat(torch.randn(n, 4, device=DEV))
    _, trap_p = proc._tables()
    x = mods["B"].p_sample_chain(net.flat_params_nograd(), proc._sched, trap_p, x0, 999, 20, seed=1, precision=1)
    x = mods["B"].p_sample_chain(net.flat_params_nograd(), proc._sched, trap_p, x, 19, 20, seed=1, precision=1)
    net.precision = "fp32"
    assert not torch.isnan(x).any()
    assert float((x @ x.transpose(-1, -2) - torch.eye(3, device=DEV)).abs().max()) < 1e-4
    assert float((torch.linalg.det(x) - 1).abs().max()) < 1e-4


def test_full_chain_is_the_same_however_it_is_cut_into_launches(mods, net):
    """BASELINE config 3 verbatim (2^20 rotations, all 1000 reverse steps, bf16): one 1000-step launch, ten 100-step launches
    and a ragged cut (1 + 333 + 666) are the same chain -- the noise is keyed by (seed, sample, rng_offset + t), the
    per-launch tables cover exactly the rows a launch runs, and nothing is carried between launches but the rotations --
    and running it twice gives the same bits.  The end state is a batch of finite rotations."""
    B = mods["B"]
    net.precision = "bf16"
    proc = mods["diff"].SO3Diffusion(net, timesteps=1000).to(DEV)
    _, trap_p = proc._tables()
    params = net.flat_params_nograd()
    net.precision = "fp32"
    n = 1 << 20
    x0 = mods["util"].quat_to_rmat(torch.randn(n, 4, device=DEV, generator=torch.Generator(device=DEV).manual_seed(11)))

    def run(cuts):
        x, t = x0, 999
        for k in cuts:
            x = B.p_sample_chain(params, proc._sched, trap_p, x, t, k, seed=5, rng_offset=17, precision=1, guide_p=proc._guide_p)
            t -= k
        assert t == -1
        return x

    whole = run([1000])
    assert torch.isfinite(whole).all()
    assert float((whole @ whole.transpose(-1, -2) - torch.eye(3, device=DEV)).abs().max()) < 1e-4
    assert float((torch.linalg.det(whole) - 1).abs().max()) < 1e-4
    assert torch.equal(run([1000]), whole)                     # deterministic
    # a cut hands the state over as a rotation matrix (the reference's format) instead of the quaternion the kernel keeps
    # inside a launch: a 1e-7 re-rounding per cut, carried through the remaining steps -- not bits, but the same chain
    for cuts in ([100] * 10, [1, 333, 666]):
        d = (run(cuts) - whole).abs().reshape(n, -1).max(1).values
        # (measured: median 4e-6, 99th percentile 1.3e-5; a handful of the 2^20 chains take another branch somewhere and end elsewhere)
        assert float(d.median()) < 1e-5 and float(d.quantile(0.99)) < 1e-3, (cuts, float(d.median()), float(d.quantile(0.99)))


# ------------------------------------------------------------------ statistics
def test_mmd_kernel_sums_vs_oracle_and_reference_samples(mods, golden):
    U = mods["util"]
    ref = golden["chain_samples_trained"]["x_final"]
    rng_ = np.random.default_rng(0)
    uni = O.quat_to_rmat(rng_.standard_normal((1500, 4)).astype(np.float32))
    X, Y = dev(ref[:2000]), dev(uni)
    mine = float(U.MMD(X, Y, U.rmat_gaussian_kernel))
    want = O.MMD(ref[:2000], uni)
    assert abs(mine - want) < 2e-6 + 1e-5 * abs(want), (mine, want)
    # the generic (broadcast) path of MMD with a user kernel agrees with the fused one
    Xs, Ys = X[:300], Y[:257]
    fused = float(U.MMD(Xs, Ys, U.rmat_gaussian_kernel))
    generic = float(U.MMD(Xs, Ys, lambda a, b: U.rmat_gaussian_kernel(a, b)))
    assert abs(fused - generic) < 1e-5
    cos_f = float(U.MMD(Xs, Ys, U.rmat_cosine_kernel))
    cos_g = float(U.MMD(Xs, Ys, lambda a, b: U.rmat_cosine_kernel(a, b)))
    assert abs(cos_f - cos_g) < 1e-5
    assert U.Ker_2samp_test(dev(ref[:2048]), dev(ref[2048:]), U.rmat_gaussian_kernel)
    assert not U.Ker_2samp_test(dev(ref[:1500]), Y, U.rmat_gaussian_kernel)
    # bingham_test.py-sized problem (20,000 x 20,000 pairs) in one launch: identical populations give MMD ~ 0
    big = mods["B"].quat_to_rmat(torch.randn(20000, 4, device=DEV))
    assert abs(float(U.MMD(big, big, U.rmat_gaussian_kernel))) < 1e-6


def test_mmd_statistics_vs_reference_values(mods, golden):
    U = mods["util"]
    g = golden["stats"]
    X, Y = dev(g["X"]), dev(g["Y"])
    assert abs(float(U.MMD(X, Y, U.rmat_gaussian_kernel)) - float(g["mmd_gauss"])) < 5e-6
    assert abs(float(U.MMD(X, Y, U.rmat_cosine_kernel)) - float(g["mmd_cos"])) < 2e-5
    assert maxabs(host(U.rmat_gaussian_kernel(X[:50].unsqueeze(0), Y[:40].unsqueeze(1))), g["kern_gauss_xy"]) < 5e-6
    assert maxabs(host(U.rmat_cosine_dist(X[:100], Y[:100])), g["cos_dist"]) < 2e-6
    assert U.Ker_2samp_test(X[:150], X[150:], U.rmat_gaussian_kernel) == bool(g["test_same"])
    assert U.Ker_2samp_test(X[:257], Y, U.rmat_gaussian_kernel) == bool(g["test_diff"])
    assert abs(U.Ker_2samp_log_prob(X[:257], Y, U.rmat_gaussian_kernel) - float(g["logp_diff"])) < 1e-3


def test_search_guide_is_bit_identical(mods):
    """the [rows][258] guide only narrows the bisection's starting bracket: angles, rotations and targets are bitwise the
    same with and without it, for explicit draws (incl. u = 0 and u just below 1) and Philox draws, per-sample rows"""
    B = mods["B"]
    T = 1000
    sched = dev(B.schedule_from_betas(O.cosine_beta_schedule(T)))
    for row in (4, 12):
        trap = B.igso3_build_tables(sched[row])
        guide = B.igso3_build_guide(trap)
        gh = host(guide).astype(np.int64) & 0xffff
        th = host(trap)
        for r in (0, 1, 500, 999):                       # guide[b] = #{k : row[k] <= b/256}
            cnt = (th[r][None, :] <= (np.arange(257, dtype=np.float32) / np.float32(256))[:, None]).sum(1)
            assert (gh[r, :257] == cnt).all()
        n = 5000
        gen = torch.Generator(device=DEV).manual_seed(row)
        ri = torch.randint(0, T, (n,), device=DEV, generator=gen)
        unif = torch.rand(n, device=DEV, generator=gen)
        unif[:4] = torch.tensor([0.0, 1.0 - 2.0 ** -24, 0.5, 2.0 ** -30], device=DEV)
        axes = torch.randn(n, 3, device=DEV, generator=gen)
        for kw in (dict(axes=axes, unif=unif), dict(seed=3, rng_offset=9)):
            a = B.igso3_sample(trap, n, row_idx=ri, quirk_col0=True, want_angle=True, **kw)
            b = B.igso3_sample(trap, n, row_idx=ri, quirk_col0=True, want_angle=True, guide=guide, **kw)
            assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    # the chain kernels with the guide of the p rows: same samples, bit for bit
    trap_p = B.igso3_build_tables(sched[12])
    guide_p = B.igso3_build_guide(trap_p)
    from so3x.so3_train import RotPredict
    torch.manual_seed(0)
    pnet = RotPredict(out_type="skewvec").to(DEV).flat_params_nograd()
    xc = mods["util"].quat_to_rmat(torch.randn(777, 4, device=DEV))
    for prec in (0, 1):
        a = B.p_sample_chain(pnet, sched, trap_p, xc, 700, 40, seed=2, precision=prec)
        b = B.p_sample_chain(pnet, sched, trap_p, xc, 700, 40, seed=2, precision=prec, guide_p=guide_p)
        assert torch.equal(a, b)
    trap_q = B.igso3_build_tables(sched[4])
    guide_q = B.igso3_build_guide(trap_q)
    x0 = mods["util"].quat_to_rmat(torch.randn(3000, 4, device=DEV))
    t = torch.randint(0, T, (3000,), device=DEV)
    a = B.q_sample_target(sched, trap_q, x0, t, seed=5, rng_offset=1)
    b = B.q_sample_target(sched, trap_q, x0, t, seed=5, rng_offset=1, guide_q=guide_q)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])


def test_training_forward_stash_equals_recompute(mods, golden, net):
    """so3x_mlp_fwd_stash + so3x_mlp_bwd(zstash) vs so3x_mlp_bwd with the forward recomputed inside: the training forward
    runs on the table-folded weight image (u = 16 z + 127.5, SiLU from the LDS table) and parks z = (u - 127.5) / 16 as f16,
    the recompute runs exp-based SiLUs on the plain image -- the two see activations that differ by the table's 1.3e-4 and
    pre-activations that differ in f16's last place, so the gradients agree to ~1e-3 of their norm, not bit for bit; the
    output matches the plain forward."""
    B = mods["B"]
    params = net.flat_params_nograd()
    for n in (1, 77, 4100):
        gen = torch.Generator(device=DEV).manual_seed(n)
        x = mods["util"].quat_to_rmat(torch.randn(n, 4, device=DEV, generator=gen))
        t = torch.randint(0, 300, (n,), device=DEV, generator=gen)
        dout = torch.randn(n, 3, device=DEV, generator=gen)
        out, zs = B.mlp_fwd_stash(params, x, t, 300)
        assert zs.numel() == (n + 31) // 32 * 17 * 1024
        ref_out = B.mlp_fwd(params, x, t, B.PREC_BF16, 300)
        assert float((out - ref_out).abs().max()) < 3e-2          # same operands; the plain forward folds the SiLU scale
        g_stash = B.mlp_bwd(params, x, t, dout, B.PREC_BF16, 300, zstash=zs)
        g_rec = B.mlp_bwd(params, x, t, dout, B.PREC_BF16, 300)
        assert torch.isfinite(g_stash).all()
        assert float((g_stash - g_rec).norm() / g_rec.norm()) < 4e-3, n
    with pytest.raises(B.So3xError):
        B.mlp_bwd(params, x, t, dout, B.PREC_F32, 300, zstash=zs)    # the stash belongs to the bf16 fused path


def test_training_step_as_a_captured_graph(mods, golden):
    """so3x.graphs.TrainStepGraph: the whole step (noising, network, loss, backward, fused Adam) replayed as one hipGraph
    gives exactly the losses and parameters of the same steps run eagerly -- fresh noise on every replay (device-resident
    Philox offset), fresh t (torch's graph-safe generator)."""
    import copy
    from so3x import rng
    from so3x.graphs import TrainStepGraph
    from so3x.so3_train import RotPredict
    from so3x.diffusion import SO3Diffusion
    torch.manual_seed(0)
    base = RotPredict(out_type="skewvec", precision="bf16").to(DEV)
    x = mods["util"].quat_to_rmat(torch.randn(2048, 4, device=DEV))
    results = []
    for mode in ("eager", "graph"):
        net = copy.deepcopy(base)
        proc = SO3Diffusion(net, timesteps=100).to(DEV)
        proc.rng_counter = torch.zeros(1, dtype=torch.int64, device=DEV)
        opt = torch.optim.Adam(net.parameters(), lr=1e-3, fused=True, capturable=True)
        rng.manual_seed(7)
        if mode == "graph":
            g = TrainStepGraph(proc, opt, x.shape, warmup=2)
            # rewind everything the warm-up and the capture touched, then replay from the same state as the eager run
            net.load_state_dict(base.state_dict())
            opt.load_state_dict(torch.optim.Adam(net.parameters(), lr=1e-3, fused=True, capturable=True).state_dict())
            proc.rng_counter.zero_()
        torch.manual_seed(11)
        torch.cuda.manual_seed(11)
        losses = []
        for _ in range(4):
            if mode == "eager":
                opt.zero_grad(set_to_none=True)
                loss = proc(x)
                loss.backward()
                opt.step()
                losses.append(float(loss.detach()))
            else:
                losses.append(float(g.step(x)))
        results.append((losses, torch.cat([p.detach().reshape(-1) for p in net.parameters()]).clone(), int(proc.rng_counter)))
    (le, pe, ce), (lg, pg, cg) = results
    assert ce == cg == 4
    assert len(set(lg)) == 4                                   # different noise / t on every replay
    assert all(np.isfinite(lg))
    # the eager and the replayed steps see the same Philox offsets; t comes from torch's generator, whose graph-safe
    # offset bookkeeping differs from the eager one, so compare statistically: same loss scale, parameters moved alike
    assert abs(np.mean(lg) - np.mean(le)) < 0.5 * max(np.mean(le), 1e-3)
    assert float((pg - pe).abs().max()) < 2e-2


@pytest.mark.gpu
def test_so3_bezier_de_casteljau(mods):
    """util.so3_bezier (reference util.py:340-346): two control points = so3_lerp; end points are interpolated; control
    points on one geodesic give the point of the same geodesic"""
    from so3x import util
    n = 500
    g = torch.Generator(device=DEV).manual_seed(9)
    a, b, c = (util.quat_to_rmat(torch.randn(n, 4, device=DEV, generator=g)) for _ in range(3))
    w = torch.rand(n, 1, device=DEV, generator=g)
    assert torch.equal(util.so3_bezier(a, b, weight=w), util.so3_lerp(a, b, w))
    zero, one = torch.zeros(n, 1, device=DEV), torch.ones(n, 1, device=DEV)
    assert float((util.so3_bezier(a, b, c, weight=zero) - a).abs().max()) < 2e-5
    assert float((util.so3_bezier(a, b, c, weight=one) - c).abs().max()) < 1e-3   # a exp(log(a^T c)): 1e-7 / (pi - angle) conditioning
    mid = util.so3_lerp(a, c, torch.full((n, 1), 0.5, device=DEV))        # a, mid, c on one geodesic
    out = util.so3_bezier(a, mid, c, weight=w)
    # (1-w)^2 * 0 + 2 w (1-w) * 1/2 + w^2 * 1 = w along that geodesic
    # (a^T c has an angle < pi almost surely, so the geodesic and its midpoint are unique)
    assert float((out - util.so3_lerp(a, c, w)).abs().max()) < 2e-3 and float((out - util.so3_lerp(a, c, w)).abs().median()) < 1e-6
    with pytest.raises(ValueError):
        util.so3_bezier(a, weight=w)


_DP_WORKER = r'''
import os, sys, torch
sys.path.insert(0, sys.argv[1])
from so3x import parallel, backend as B
from so3x.so3_train import RotPredict
from so3x.diffusion import SO3Diffusion
ctx = parallel.init()                       # SO3X_DIST_BACKEND=gloo, both ranks on cuda:0
assert ctx.world_size == 2 and ctx.device.type == "cuda"
torch.manual_seed(ctx.rank)                 # different initial weights per rank on purpose
net = RotPredict(out_type="skewvec", precision="bf16").to(ctx.device)
parallel.broadcast_parameters(net, ctx)
proc = SO3Diffusion(net, timesteps=100).to(ctx.device)
opt = torch.optim.Adam(net.parameters(), lr=1e-3, fused=True)
glob = 2048
lo, hi = parallel.shard_range(glob, ctx.rank, ctx.world_size)
proc.index_base = lo
x_all = B.quat_to_rmat(torch.randn(glob, 4, generator=torch.Generator().manual_seed(7)).to(ctx.device))
losses = []
for step in range(5):
    loss = proc(x_all[lo:hi])
    opt.zero_grad()
    loss.backward()
    parallel.allreduce_gradients(net, ctx)
    opt.step()
    losses.append(parallel.mean_scalar(loss.detach(), ctx))
flat = torch.cat([p.data.reshape(-1) for p in net.parameters()])
both = [torch.zeros_like(flat) for _ in range(2)]
torch.distributed.all_gather(both, flat)
assert torch.equal(both[0], both[1]), "replicas diverged"
assert all(l == l and l < 1e3 for l in losses)
parallel.finalize(ctx)
print("OK", ctx.rank, losses[0], losses[-1])
'''


@pytest.mark.gpu
def test_data_parallel_training_two_ranks_on_one_gpu(tmp_path):
    """the DP training plumbing (broadcast, sharded batch with global Philox indices, one flat gradient all-reduce, Adam)
    with real device tensors: two ranks share cuda:0, collectives over gloo; replicas must stay bit-identical"""
    import os, subprocess, sys
    from conftest import PKG
    script = tmp_path / "dp_worker.py"
    script.write_text(_DP_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29547", WORLD_SIZE="2", LOCAL_RANK="0",
               SO3X_DIST_BACKEND="gloo")
    procs = [subprocess.Popen([sys.executable, str(script), PKG], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=600)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0 and "OK" in o, o


@pytest.mark.gpu
def test_captured_graph_survives_bigger_eager_calls_on_other_streams(mods):
    """the binding's scratch buffers are per (device, stream) and a graph's buffer comes from the graph's own pool: eager
    calls that grow the scratch elsewhere between replays must not disturb the captured training step"""
    from so3x import rng
    from so3x.so3_train import RotPredict
    from so3x.graphs import TrainStepGraph
    B = mods["B"]
    x = B.quat_to_rmat(torch.randn(512, 4, device=DEV, generator=torch.Generator(device=DEV).manual_seed(1)))
    big_x = B.quat_to_rmat(torch.randn(1 << 17, 4, device=DEV, generator=torch.Generator(device=DEV).manual_seed(2)))
    big_t = torch.randint(0, 100, (1 << 17,), device=DEV, generator=torch.Generator(device=DEV).manual_seed(3))
    big_d = torch.ones(1 << 17, 3, device=DEV)

    def run(disturb):
        torch.manual_seed(0)
        torch.cuda.manual_seed(0)
        rng.manual_seed(9)
        net = RotPredict(out_type="skewvec", precision="bf16").to(DEV)
        proc = mods["diff"].SO3Diffusion(net, timesteps=100).to(DEV)
        opt = torch.optim.Adam(net.parameters(), lr=1e-3, fused=True, capturable=True)
        g = TrainStepGraph(proc, opt, x.shape, warmup=2)
        losses = []
        for i in range(4):
            if disturb:  # fp32 staged backward at 2^17 samples: a scratch request hundreds of times the graph's
                B.mlp_bwd(net.flat_params_nograd(), big_x, big_t, big_d, B.PREC_F32, 100)
                B.mlp_fwd(net.flat_params_nograd(), big_x, big_t, B.PREC_BF16, 100)
            losses.append(float(g.step(x)))
        return losses, torch.cat([p.detach().reshape(-1) for p in net.parameters()]).clone()

    l0, p0 = run(False)
    l1, p1 = run(True)
    assert l0 == l1 and torch.equal(p0, p1)


@pytest.mark.gpu
@pytest.mark.parametrize("n", [1, 63, 65, 255, 257, 1000])
@pytest.mark.parametrize("pad", [64, 3])   # 64 floats: outputs stay 16-byte aligned; 3: deliberately misaligned
def test_outputs_stay_inside_their_buffers(mods, n, pad):
    """no GPU AddressSanitizer on this pool: instead every output of the tile-staged kernels is placed between two guard
    bands of sentinels (through the raw C ABI) and the bands must come back untouched, for ragged sizes around the tile and
    wave boundaries and for aligned and misaligned buffers"""
    import ctypes as C
    B = mods["B"]
    lib = B.lib()
    G = 256
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    SENT = 12345.0

    def guarded(count):
        buf = torch.full((count + 2 * G + pad,), SENT, device=DEV)
        return buf, buf[G + pad: G + pad + count]

    def check(buf, count, what):
        torch.cuda.synchronize()
        assert bool((buf[:G + pad] == SENT).all()) and bool((buf[G + pad + count:] == SENT).all()), f"{what}: wrote outside its output"
        assert not bool((buf[G + pad: G + pad + count] == SENT).any()), f"{what}: left output unwritten"

    p = lambda t: C.c_void_p(t.data_ptr())
    q = torch.randn(n, 4, device=DEV)
    R = B.quat_to_rmat(q)
    R2 = B.quat_to_rmat(torch.randn(n, 4, device=DEV))
    k = torch.rand(n, device=DEV)
    x6 = torch.randn(n, 6, device=DEV)
    Gm = torch.randn(n, 3, 3, device=DEV)
    cases = [
        ("quat_to_rmat", 9, lambda o: lib.so3x_quat_to_rmat(st, p(q), p(o), C.c_int64(n))),
        ("so3_scale", 9, lambda o: lib.so3x_so3_scale(st, p(R), p(k), C.c_int64(1), p(o), C.c_int64(n))),
        ("log_rmat_vec", 3, lambda o: lib.so3x_log_rmat_vec(st, p(R), p(o), C.c_int64(n))),
        ("log_rmat", 9, lambda o: lib.so3x_log_rmat(st, p(R), p(o), C.c_int64(n))),
        ("rmat_dist", 1, lambda o: lib.so3x_rmat_dist(st, p(R), p(R2), p(o), C.c_int64(n))),
        ("rmul", 9, lambda o: lib.so3x_rmul(st, p(R), C.c_int64(9), p(R2), C.c_int64(9), C.c_int(1), p(o), C.c_int64(n))),
        ("six2rmat", 9, lambda o: lib.so3x_six2rmat(st, p(x6), p(o), C.c_int64(n))),
        ("six2rmat_bwd", 6, lambda o: lib.so3x_six2rmat_bwd(st, p(x6), p(Gm), p(o), C.c_int64(n))),
        ("log_rmat_bwd", 9, lambda o: lib.so3x_log_rmat_bwd(st, p(R), p(Gm), p(o), C.c_int64(n))),
    ]
    for name, w, call in cases:
        buf, out = guarded(n * w)
        assert call(out) == 0, name
        check(buf, n * w, name)
    # two outputs at once
    bufa, oa = guarded(n * 9)
    bufb, ob = guarded(n * 9)
    assert lib.so3x_rmat_dist_bwd(st, p(R), p(R2), p(k), p(oa), p(ob), C.c_int64(n)) == 0
    check(bufa, n * 9, "rmat_dist_bwd.da"); check(bufb, n * 9, "rmat_dist_bwd.db")
    bufl, ol = guarded(n)
    bufs, os_ = guarded(n * 3)
    assert lib.so3x_igso3_logprob_score(st, p(R), p(k * 0.8 + 0.2), C.c_int64(1), p(ol), p(os_), None, C.c_int64(n)) == 0
    check(bufl, n, "logprob"); check(bufs, n * 3, "score")
    # the network forward (n_out = 3 and 6) and the chain's x_out
    net3 = mods["train"].RotPredict(out_type="skewvec", precision="bf16").to(DEV)
    net6 = mods["train"].RotPredict(out_type="rotmat", precision="bf16").to(DEV)
    t = torch.randint(0, 50, (n,), device=DEV)
    for net_, no in ((net3, 3), (net6, 6)):
        prm = net_.flat_params_nograd()
        nb = lib.so3x_mlp_workspace_bytes(C.c_int64(0), C.c_int(1), C.c_int(50))
        ws = torch.empty(nb, dtype=torch.uint8, device=DEV)
        buf, out = guarded(n * no)
        assert lib.so3x_mlp_fwd(st, p(prm), p(R), p(t), C.c_int64(1), p(out), C.c_int64(n), C.c_int(no), C.c_int(1), C.c_int(50),
                                p(ws), C.c_size_t(nb)) == 0
        check(buf, n * no, f"mlp_fwd n_out={no}")


@pytest.mark.gpu
@pytest.mark.parametrize("n", [1, 33, 257, 1000])
def test_network_and_diffusion_outputs_stay_inside_their_buffers(mods, n):
    """guard bands (see above) around the outputs of the fused kernels: noising, both chains, both networks' gradients"""
    import ctypes as C
    from so3x.so3_lock_train import RotPredict as Wide
    B = mods["B"]
    lib = B.lib()
    G, SENT = 256, 12345.0
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None

    def guarded(count):
        buf = torch.full((count + 2 * G,), SENT, device=DEV)
        return buf, buf[G: G + count]

    def check(buf, count, what):
        torch.cuda.synchronize()
        assert bool((buf[:G] == SENT).all()) and bool((buf[G + count:] == SENT).all()), f"{what}: wrote outside its output"
        assert not bool((buf[G: G + count] == SENT).any()), f"{what}: left output unwritten"

    T = 50
    net = mods["train"].RotPredict(out_type="skewvec", precision="bf16").to(DEV)
    proc = mods["diff"].SO3Diffusion(net, timesteps=T).to(DEV)
    trap_q, trap_p = proc._tables()
    R = B.quat_to_rmat(torch.randn(n, 4, device=DEV))
    t = torch.randint(0, T, (n,), device=DEV)
    bx, ox = guarded(n * 9)
    bt, ot = guarded(n * 3)
    assert lib.so3x_q_sample_target(st, p(proc._sched), C.c_int(T), p(trap_q), p(proc._guide_q), p(R), p(t), C.c_int(1), None, None,
                                    None, C.c_uint64(1), C.c_uint64(0), None, C.c_int64(0), p(ox), p(ot), None, C.c_int64(n)) == 0
    check(bx, n * 9, "q_sample x_t"); check(bt, n * 3, "q_sample target")
    for wide in (False, True):
        nt = Wide(out_type="skewvec", precision="bf16").to(DEV) if wide else net
        prm = nt.flat_params_nograd()
        fn = lib.so3x_resnet_p_sample_chain if wide else lib.so3x_p_sample_chain
        nb = lib.so3x_resnet_workspace_bytes(C.c_int(1), C.c_int(T)) if wide else lib.so3x_p_sample_workspace_bytes(C.c_int(T), C.c_int(1))
        ws = torch.empty(nb, dtype=torch.uint8, device=DEV)
        bo, oo = guarded(n * 9)
        assert fn(st, p(prm), p(proc._sched), C.c_int(T), p(trap_p), p(proc._guide_p), p(R), p(oo), C.c_int(T - 1), C.c_int(3), None, None,
                  C.c_uint64(1), C.c_uint64(0), C.c_int64(0), C.c_int64(n), C.c_int(1), p(ws), C.c_size_t(nb)) == 0
        check(bo, n * 9, "wide chain x_out" if wide else "chain x_out")
        # gradients: exactly nparams floats
        for prec in ((1, 0) if not wide else (1,)):
            dout = torch.randn(n, 3, device=DEV)
            npar = prm.numel()
            bg, og = guarded(npar)
            if wide:
                nbw = lib.so3x_resnet_train_workspace_bytes(C.c_int64(n), C.c_int(prec), C.c_int(T))
                wsb = torch.empty(nbw, dtype=torch.uint8, device=DEV)
                rc = lib.so3x_resnet_bwd(st, p(prm), p(R), p(t), C.c_int64(1), p(dout), p(og), C.c_int64(n), C.c_int(3), C.c_int(prec),
                                         C.c_int(T), None, p(wsb), C.c_size_t(nbw))
            else:
                nbw = lib.so3x_mlp_workspace_bytes(C.c_int64(n), C.c_int(prec), C.c_int(T))
                wsb = torch.empty(nbw, dtype=torch.uint8, device=DEV)
                rc = lib.so3x_mlp_bwd(st, p(prm), p(R), p(t), C.c_int64(1), p(dout), p(og), C.c_int64(n), C.c_int(3), C.c_int(prec),
                                      C.c_int(T), None, p(wsb), C.c_size_t(nbw))
            assert rc == 0
            check(bg, npar, f"{'wide' if wide else 'mlp'} bwd dparams prec={prec}")


@pytest.mark.gpu
@pytest.mark.parametrize("n", [33, 1000, 70001])
def test_workspaces_are_big_enough_as_advertised(mods, n):
    """every *_workspace_bytes / *_stash_bytes figure is an upper bound on what the kernels touch: the scratch is placed
    between guard bands at exactly the advertised size"""
    import ctypes as C
    from so3x.so3_lock_train import RotPredict as Wide
    B = mods["B"]
    lib = B.lib()
    GB = 4096
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None

    def guarded(nbytes):
        buf = torch.full((nbytes + 2 * GB,), 0xA5, dtype=torch.uint8, device=DEV)
        return buf, buf[GB: GB + nbytes]

    def check(buf, nbytes, what):
        torch.cuda.synchronize()
        assert bool((buf[:GB] == 0xA5).all()) and bool((buf[GB + nbytes:] == 0xA5).all()), f"{what}: wrote outside its scratch"

    T = 50
    R = B.quat_to_rmat(torch.randn(n, 4, device=DEV))
    t = torch.randint(0, T, (n,), device=DEV)
    dout = torch.randn(n, 3, device=DEV)
    out = torch.empty(n, 3, device=DEV)
    net = mods["train"].RotPredict(out_type="skewvec", precision="bf16").to(DEV)
    prm = net.flat_params_nograd()
    dprm = torch.empty_like(prm)
    for prec in (1, 0):
        for tt in (T, 0):
            nb = lib.so3x_mlp_workspace_bytes(C.c_int64(n), C.c_int(prec), C.c_int(tt))
            buf, ws = guarded(nb)
            assert lib.so3x_mlp_bwd(st, p(prm), p(R), p(t), C.c_int64(1), p(dout), p(dprm), C.c_int64(n), C.c_int(3), C.c_int(prec), C.c_int(tt),
                                    None, p(ws), C.c_size_t(nb)) == 0
            check(buf, nb, f"mlp_bwd prec={prec} t_table={tt}")
            nbf = lib.so3x_mlp_workspace_bytes(C.c_int64(0), C.c_int(prec), C.c_int(tt))
            buf, ws = guarded(nbf)
            assert lib.so3x_mlp_fwd(st, p(prm), p(R), p(t), C.c_int64(1), p(out), C.c_int64(n), C.c_int(3), C.c_int(prec), C.c_int(tt),
                                    p(ws), C.c_size_t(nbf)) == 0
            check(buf, nbf, f"mlp_fwd prec={prec} t_table={tt}")
    # stash forward: both the scratch and the stash itself
    nbf = lib.so3x_mlp_workspace_bytes(C.c_int64(0), C.c_int(1), C.c_int(T))
    nbs = lib.so3x_mlp_stash_bytes(C.c_int64(n))
    bufw, ws = guarded(nbf)
    bufs, zs = guarded(nbs)
    assert lib.so3x_mlp_fwd_stash(st, p(prm), p(R), p(t), C.c_int64(1), p(out), p(zs), C.c_int64(n), C.c_int(3), C.c_int(1), C.c_int(T),
                                  p(ws), C.c_size_t(nbf)) == 0
    check(bufw, nbf, "mlp_fwd_stash scratch"); check(bufs, nbs, "mlp_fwd_stash stash")
    # the chain's scratch (weight image + per-timestep tables), both networks
    proc = mods["diff"].SO3Diffusion(net, timesteps=T).to(DEV)
    _, trap_p = proc._tables()
    xo = torch.empty_like(R)
    for prec in (1, 0):
        nbc = lib.so3x_p_sample_workspace_bytes(C.c_int(T), C.c_int(prec))
        buf, ws = guarded(nbc)
        assert lib.so3x_p_sample_chain(st, p(prm), p(proc._sched), C.c_int(T), p(trap_p), p(proc._guide_p), p(R), p(xo), C.c_int(T - 1),
                                       C.c_int(2), None, None, C.c_uint64(1), C.c_uint64(0), C.c_int64(0), C.c_int64(n), C.c_int(prec),
                                       p(ws), C.c_size_t(nbc)) == 0
        check(buf, nbc, f"p_sample_chain scratch prec={prec}")
    # the wide network: forward scratch, training scratch, stash
    wide = Wide(out_type="skewvec", precision="bf16").to(DEV)
    wprm = wide.flat_params_nograd()
    wd = torch.empty_like(wprm)
    for prec in (1, 0):
        nbw = lib.so3x_resnet_workspace_bytes(C.c_int(prec), C.c_int(T))
        nbst = lib.so3x_resnet_stash_bytes(C.c_int64(n), C.c_int(prec))
        bufw, ws = guarded(nbw)
        bufs, stash = guarded(nbst)
        assert lib.so3x_resnet_fwd_stash(st, p(wprm), p(R), p(t), C.c_int64(1), p(out), p(stash), C.c_int64(n), C.c_int(3), C.c_int(prec),
                                         C.c_int(T), p(ws), C.c_size_t(nbw)) == 0
        check(bufw, nbw, f"resnet_fwd_stash scratch prec={prec}"); check(bufs, nbst, f"resnet stash prec={prec}")
        nbt = lib.so3x_resnet_train_workspace_bytes(C.c_int64(n), C.c_int(prec), C.c_int(T))
        buft, wst = guarded(nbt)
        assert lib.so3x_resnet_bwd(st, p(wprm), p(R), p(t), C.c_int64(1), p(dout), p(wd), C.c_int64(n), C.c_int(3), C.c_int(prec), C.c_int(T),
                                   p(stash), p(wst), C.c_size_t(nbt)) == 0
        check(buft, nbt, f"resnet_bwd scratch (with stash) prec={prec}")
        buft, wst = guarded(nbt)
        assert lib.so3x_resnet_bwd(st, p(wprm), p(R), p(t), C.c_int64(1), p(dout), p(wd), C.c_int64(n), C.c_int(3), C.c_int(prec), C.c_int(T),
                                   None, p(wst), C.c_size_t(nbt)) == 0
        check(buft, nbt, f"resnet_bwd scratch (recompute) prec={prec}")
        nbw = lib.so3x_resnet_workspace_bytes(C.c_int(prec), C.c_int(T))
        buf, ws = guarded(nbw)
        assert lib.so3x_resnet_p_sample_chain(st, p(wprm), p(proc._sched), C.c_int(T), p(trap_p), p(proc._guide_p), p(R), p(xo),
                                              C.c_int(T - 1), C.c_int(2), None, None, C.c_uint64(1), C.c_uint64(0), C.c_int64(0),
                                              C.c_int64(n), C.c_int(prec), p(ws), C.c_size_t(nbw)) == 0
        check(buf, nbw, f"resnet chain scratch prec={prec}")


_SHARD_WORKER = r'''
import os, sys, torch
sys.path.insert(0, sys.argv[1])
import so3x
from so3x import parallel
from so3x.so3_train import RotPredict
from so3x.diffusion import SO3Diffusion
ctx = parallel.init()
torch.manual_seed(3)
net = RotPredict(out_type="skewvec", precision="bf16").to(ctx.device)
proc = SO3Diffusion(net, timesteps=30).to(ctx.device)
so3x.manual_seed(5)
x = parallel.sharded_p_sample_loop(proc, 1001, ctx, gather=True)
assert x.shape == (1001, 3, 3)
if ctx.rank == 0:
    torch.save(x.cpu(), sys.argv[2])
parallel.finalize(ctx)
print("OK", ctx.rank)
'''


@pytest.mark.gpu
def test_sharded_sampling_is_invariant_to_the_number_of_ranks(tmp_path):
    """parallel.sharded_p_sample_loop: two ranks (sharing cuda:0, gloo) assemble bit for bit the tensor one rank produces --
    Philox counters are keyed by the global sample index, the chain has no collective"""
    import os, subprocess, sys
    import so3x
    from conftest import PKG
    from so3x import parallel
    from so3x.so3_train import RotPredict
    from so3x.diffusion import SO3Diffusion
    script = tmp_path / "shard_worker.py"
    script.write_text(_SHARD_WORKER)
    outp = str(tmp_path / "two_ranks.pt")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), WORLD_SIZE="2", LOCAL_RANK="0", SO3X_DIST_BACKEND="gloo")
    procs = [subprocess.Popen([sys.executable, str(script), PKG, outp], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=600)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0 and "OK" in o, o
    two = torch.load(outp)
    torch.manual_seed(3)
    net = RotPredict(out_type="skewvec", precision="bf16").to(DEV)
    proc = SO3Diffusion(net, timesteps=30).to(DEV)
    so3x.manual_seed(5)
    one = parallel.sharded_p_sample_loop(proc, 1001, parallel.Ctx(0, 1, 0, torch.device(DEV), False))
    assert torch.equal(one.cpu(), two)
    assert proc.index_base == 0


@pytest.mark.parametrize("wrong", ["sigma_t x 1.15", "sigma_t x 0.87", "network output x 0.9", "network output x 1.1"])
def test_G3_has_power_against_near_misses(mods, golden, net, wrong):
    """Haar-uniform is the only alternative G3 was shown to reject.  Near misses -- the posterior noise scale
    off by 15 %, the score network's output off by 10 % -- leave the kernel MMD far inside its acceptance bound (measured
    4e-4 .. 8e-4 against a bound of 7.6e-2: the bound of util.py:289-299 is a worst-case one), so the chain-level gate gets a
    second, sharper statistic: the median geodesic distance to the nearer training mode (so3_train.py:65-72).  The shipped bf16
    chain reproduces the reference population's median within 6 % (measured 0.5-1.6 %; the bootstrap standard error of the
    reference's own median is below 2 %); every near miss moves it by more than that, in the direction the error implies."""
    import copy
    from so3x import rng
    g = golden["chain_samples_trained"]
    ref = g["x_final"].astype(np.float64)
    m = len(ref)
    z90 = np.array([[0.0, -1.0, 0.0], [1.0, 0.0, 0.0], [0.0, 0.0, 1.0]])

    def median_mode_distance(X):
        d0 = O.rmat_dist(X, np.broadcast_to(z90, X.shape).copy(), "f64")
        d1 = O.rmat_dist(X, np.broadcast_to(z90.T, X.shape).copy(), "f64")
        return float(np.median(np.minimum(d0, d1)))

    def chain(sig=1.0, vscale=1.0):
        n2 = copy.deepcopy(net)
        n2.load_state_dict({f"net.{l}.{k}": torch.from_numpy(g[f"net_{l}_{k}"]) for l in (0, 2, 4, 6, 8) for k in ("weight", "bias")})
        n2 = n2.to(DEV)
        n2.precision = "bf16"
        with torch.no_grad():
            n2.net[8].weight.mul_(vscale)
            n2.net[8].bias.mul_(vscale)
        proc = mods["diff"].SO3Diffusion(n2, timesteps=1000).to(DEV)
        if sig != 1.0:
            proc._tables()
            proc._sched[12] *= sig
            proc._trap_p = mods["B"].igso3_build_tables(proc._sched[12])
            proc._guide_p = mods["B"].igso3_build_guide(proc._trap_p)
        rng.manual_seed(2024)
        return host(proc.p_sample_loop((m,))).astype(np.float64)

    med_ref = median_mode_distance(ref)
    boot = np.random.default_rng(0)
    se = np.std([median_mode_distance(ref[boot.integers(0, m, m)]) for _ in range(40)]) / med_ref
    assert se < 0.02, se                                                    # 6 % is at least three standard errors
    shipped = median_mode_distance(chain())
    assert abs(shipped - med_ref) < 0.06 * med_ref, (shipped, med_ref)
    sig, vs = {"sigma_t x 1.15": (1.15, 1.0), "sigma_t x 0.87": (0.87, 1.0), "network output x 0.9": (1.0, 0.9), "network output x 1.1": (1.0, 1.1)}[wrong]
    x = chain(sig, vs)
    med = median_mode_distance(x)
    wider = sig > 1.0 or vs < 1.0          # more noise, or a weaker pull towards the data: a wider population
    assert (med - med_ref) * (1 if wider else -1) > 0.06 * med_ref, (wrong, med, med_ref)
    assert O.MMD(x, ref) > 1.5 * O.MMD(chain(), ref)                       # the kernel statistic moves the same way, inside its loose bound
