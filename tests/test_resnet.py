"""The wide residual score network of so3_lock_train.py:11-59: oracle vs the values captured
from the reference (CPU), device kernels vs the oracle through the C ABI (GPU)."""
import numpy as np
import pytest
import torch

from oracle import oracle as O

DEV = "cuda:0"


def dev(a, dtype=torch.float32):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV).to(dtype)


def host(t):
    return t.detach().cpu().numpy()


# ----------------------------------------------------------------------------- CPU: oracle pinned by the reference
def test_oracle_resnet_forward_vs_reference(golden):
    g = golden["resnet"]
    assert g["params"].size == O.N_PARAMS_RESNET
    assert [str(s) for s in g["param_names"][:2]] == ["net.0.layer.0.weight", "net.0.layer.0.bias"]
    o32 = O.resnet_fwd(g["params"], g["x"], g["t"], "f32")
    o64 = O.resnet_fwd(g["params"], g["x"], g["t"], "f64")
    assert np.abs(o32 - g["out"]).max() < 5e-6          # fp32 restatement vs the reference's fp32 forward
    assert np.abs(o64 - g["out_64"]).max() < 2e-7       # fp64 vs the reference run in double (fp32 embedding angles)
    # (1,)-shaped t broadcasts to the batch (so3_lock_train.py:52-53)
    assert np.abs(O.resnet_fwd(g["params"], g["x"], g["t"][:1], "f32") - g["out_t1"]).max() < 5e-6


def test_oracle_resnet_backward_vs_reference_autograd(golden):
    g = golden["resnet"]
    B = g["x"].shape[0]
    out = O.resnet_fwd(g["params"], g["x"], g["t"], "f64")
    assert abs(np.mean((out - g["target"]) ** 2) - float(g["loss"])) < 1e-6
    dout = (2.0 / (3 * B)) * (out - g["target"])         # d mse / d out
    dp = O.resnet_bwd(g["params"], g["x"], g["t"], dout, "f64")
    assert np.abs(dp - g["grad"]).max() < 2e-6 * max(1.0, np.abs(g["grad"]).max())


def test_oracle_lock_train_data_path(golden):
    """so3_lock_train.py:76-81: the arc so3_lerp(R_1, R_2, w) between two Euler rotations."""
    g = golden["resnet"]
    n = g["lerp_w"].shape[0]
    R1 = np.repeat(g["R1"], n, 0)
    R2 = np.repeat(g["R2"], n, 0)
    assert np.abs(O.so3_lerp(R1, R2, g["lerp_w"][:, 0], "f64") - g["lerp"]).max() < 1e-5


# ----------------------------------------------------------------------------- GPU
@pytest.fixture(scope="module")
def B():
    from so3x import backend
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return backend


@pytest.mark.gpu
def test_lock_train_data_path_on_device(B, golden):
    """so3_lock_train.py:76-81 as the script calls it: euler_to_rmat end points of shape [1,3,3], weights [B,1]"""
    from so3x.util import so3_lerp, euler_to_rmat
    from math import pi
    g = golden["resnet"]
    R1 = euler_to_rmat(torch.tensor(0.0), torch.tensor(pi / 3), torch.tensor(0.0))[None].to(DEV)
    R2 = euler_to_rmat(torch.tensor(0.0), torch.tensor(2 * pi / 3), torch.tensor(0.0))[None].to(DEV)
    assert float((R1 - dev(g["R1"])).abs().max()) < 1e-6 and float((R2 - dev(g["R2"])).abs().max()) < 1e-6
    out = so3_lerp(R1, R2, dev(g["lerp_w"]))
    assert out.shape == (32, 3, 3)
    assert float((out - dev(g["lerp"])).abs().max()) < 1e-5


@pytest.mark.gpu
def test_resnet_fwd_fp32_vs_reference_and_oracle(B, golden):
    g = golden["resnet"]
    out = host(B.resnet_fwd(dev(g["params"]), dev(g["x"]), dev(g["t"], torch.int64), 1000, precision=0))
    assert np.abs(out - g["out_64"]).max() < 2e-5        # gate G5: fp32 MFMA path vs the reference in double
    assert np.abs(out - g["out"]).max() < 2e-5
    out1 = host(B.resnet_fwd(dev(g["params"]), dev(g["x"]), dev(g["t"][:1], torch.int64), 1000, precision=0))
    assert np.abs(out1 - g["out_t1"]).max() < 2e-5


@pytest.mark.gpu
@pytest.mark.parametrize("prec,tol", [(0, 3e-5), (1, 4e-2)])
@pytest.mark.parametrize("n", [1, 31, 257, 5000])
def test_resnet_fwd_ragged_sizes_vs_oracle(B, golden, prec, tol, n):
    """every wave slot / several workgroup passes / ragged tails; bf16 operands with an fp32 residual stream"""
    g = golden["resnet"]
    rs = np.random.default_rng(n)
    x = O.quat_to_rmat(rs.standard_normal((n, 4)).astype(np.float32))
    t = rs.integers(0, 1000, n)
    ref = O.resnet_fwd(g["params"], x, t, "f64")
    out = host(B.resnet_fwd(dev(g["params"]), dev(x), dev(t, torch.int64), 1000, precision=prec))
    err = np.abs(out - ref)
    assert err.max() < tol, (n, err.max())
    if prec == 1 and n >= 257:
        assert np.median(err) < 6e-3
        col = np.arange(n) % 256 // 32                   # wave of the workgroup: early and late SIMD partners must agree in quality
        assert abs(np.median(err[col < 4]) - np.median(err[col >= 4])) < 3e-3


@pytest.mark.gpu
def test_resnet_fwd_argument_errors(B, golden):
    g = golden["resnet"]
    x = dev(g["x"])
    with pytest.raises(ValueError):
        B.resnet_fwd(dev(g["params"][:-1]), x, dev(g["t"], torch.int64), 1000)
    with pytest.raises(B.So3xError):
        B.resnet_fwd(dev(g["params"]), x, dev(g["t"], torch.int64), 0)      # a timestep table is required
    empty = B.resnet_fwd(dev(g["params"]), x[:0], dev(g["t"][:0], torch.int64), 10)
    assert empty.shape == (0, 3)


@pytest.mark.gpu
@pytest.mark.parametrize("prec", [0, 1])
def test_resnet_chain_step_vs_oracle(B, golden, prec):
    """one reverse step with the wide network vs the fp64 oracle fed the same Philox noise"""
    g = golden["resnet"]
    T = 1000
    betas = O.cosine_beta_schedule(T)
    sched = O.schedule_from_betas(betas)
    sched_d = dev(B.schedule_from_betas(betas))          # [13][T]: the 12 reference buffers + sigma_t
    trap_p = B.igso3_build_tables(sched_d[12])
    n = 300
    for t in (3, 400, 800):
        x0 = O.quat_to_rmat(np.random.default_rng(t).standard_normal((n, 4)).astype(np.float32))
        out = host(B.resnet_p_sample_chain(dev(g["params"]), sched_d, trap_p, dev(x0), t, 1, seed=4, rng_offset=20, precision=prec))
        coef = [float(sched[i][t]) for i in (6, 7, 10, 11)]
        v = O.resnet_fwd(g["params"], x0, np.full(n, t), "f64")
        x0h, ref = O.p_mean(x0, v, *coef, "f64")
        _, ang, ax = B.igso3_sample(trap_p, n, row_const=t, seed=4, rng_offset=20 + t, want_angle=True, want_axis=True)
        ref = O.rmul(ref, O.aa_to_rmat(host(ax), host(ang), "f64"), "f64")
        _, a1 = O.rmat_to_aa(x0, "f64")
        _, a2 = O.rmat_to_aa(x0h, "f64")
        cond = max(coef[0], 1.0) * (1.0 / (np.pi - a1[:, 0]) + 1.0 / (np.pi - a2[:, 0]) + 1.0)
        err = np.abs(out - ref).reshape(n, -1).max(1)
        if prec == 0:
            assert (err <= 3e-5 + 4e-6 * cond).all(), (t, float(err.max()))
        else:
            assert np.median(err) < 1e-2 * max(1.0, coef[1]), (t, float(np.median(err)))
        assert np.abs(out @ out.transpose(0, 2, 1) - np.eye(3)).max() < 1e-4


@pytest.mark.gpu
@pytest.mark.parametrize("t", [950, 998, 999])
def test_resnet_chain_bf16_step_vs_oracle_at_the_head_of_the_chain(B, golden, t):
    """the wide network's bf16 chain kernel at t >= 950 (hardware sine / cosine on 1e4-revolution arguments), step-wise against the
    f64 oracle with the tolerance derived in conftest.reverse_step_bound (the 65-wide twin of this test is
    tests/test_gpu_parity.py::test_chain_bf16_step_vs_oracle_at_the_head_of_the_chain).  dv = 3e-2: bf16 operands through the
    seven 255-wide layers (measured ~1e-2 absolute on |v| ~ 0.5)."""
    from conftest import reverse_step_bound
    g = golden["resnet"]
    T = 1000
    betas = O.cosine_beta_schedule(T)
    sched = O.schedule_from_betas(betas)
    sched_d = dev(B.schedule_from_betas(betas))
    trap_p = B.igso3_build_tables(sched_d[12])
    n = 300
    x0 = O.quat_to_rmat(np.random.default_rng(t).standard_normal((n, 4)).astype(np.float32))
    coef = [float(sched[i][t]) for i in (6, 7, 10, 11)]
    v = O.resnet_fwd(g["params"], x0, np.full(n, t), "f64")
    x0h, ref = O.p_mean(x0, v, *coef, "f64")
    _, ang, ax = B.igso3_sample(trap_p, n, row_const=t, seed=4, rng_offset=20 + t, want_angle=True, want_axis=True)
    ref = O.rmul(ref, O.aa_to_rmat(host(ax), host(ang), "f64"), "f64")
    _, a1 = O.rmat_to_aa(x0, "f64")
    _, a2 = O.rmat_to_aa(x0h, "f64")
    for prec, dv in ((1, 3e-2), (0, 2e-6)):
        out = host(B.resnet_p_sample_chain(dev(g["params"]), sched_d, trap_p, dev(x0), t, 1, seed=4, rng_offset=20, precision=prec))
        err = np.abs(out - ref).reshape(n, -1).max(1)
        bound = reverse_step_bound(coef, a1[:, 0], a2[:, 0], dv=dv)
        assert np.isfinite(out).all() and (err <= bound).all(), (t, prec, float(err.max()), float(bound[np.argmax(err - bound)]))
        assert np.abs(out @ out.transpose(0, 2, 1) - np.eye(3)).max() < 1e-4


@pytest.mark.gpu
@pytest.mark.parametrize("prec", [0, 1])
def test_resnet_chain_span_reproducible_and_shard_invariant(B, golden, prec):
    g = golden["resnet"]
    T = 40
    sched_d = dev(B.schedule_from_betas(O.cosine_beta_schedule(T)))
    trap_p = B.igso3_build_tables(sched_d[12])
    n = 700
    x0 = dev(O.quat_to_rmat(np.random.default_rng(2).standard_normal((n, 4)).astype(np.float32)))
    p = dev(g["params"])
    full = B.resnet_p_sample_chain(p, sched_d, trap_p, x0, T - 1, T, seed=6, rng_offset=0, precision=prec)
    again = B.resnet_p_sample_chain(p, sched_d, trap_p, x0, T - 1, T, seed=6, rng_offset=0, precision=prec)
    assert torch.equal(full, again)
    a = B.resnet_p_sample_chain(p, sched_d, trap_p, x0[:300], T - 1, T, seed=6, rng_offset=0, index_base=0, precision=prec)
    b = B.resnet_p_sample_chain(p, sched_d, trap_p, x0[300:], T - 1, T, seed=6, rng_offset=0, index_base=300, precision=prec)
    assert torch.equal(torch.cat([a, b]), full)          # results do not depend on how the batch is sharded over GPUs
    assert not torch.isnan(full).any()
    assert float((full @ full.transpose(-1, -2) - torch.eye(3, device=DEV)).abs().max()) < 1e-4
    # a span in one launch == the same steps one launch each, below the ill-conditioned head of the chain
    span = B.resnet_p_sample_chain(p, sched_d, trap_p, x0, 20, 21, seed=6, rng_offset=0, precision=prec)
    x = x0
    for t in reversed(range(21)):
        x = B.resnet_p_sample_chain(p, sched_d, trap_p, x, t, 1, seed=6, rng_offset=0, precision=prec)
    assert float((x - span).abs().max()) < (1e-4 if prec == 0 else 3e-2)


@pytest.mark.gpu
@pytest.mark.parametrize("n", [96, 700])
def test_resnet_backward_vs_oracle(B, golden, n):
    """fused gradient (bf16 operands, fp32 accumulation) vs the fp64 oracle; at n = 96 also vs the reference's autograd"""
    g = golden["resnet"]
    rs = np.random.default_rng(n)
    if n == 96:
        x, t, tgt = g["x"], g["t"], g["target"]
    else:
        x = O.quat_to_rmat(rs.standard_normal((n, 4)).astype(np.float32))
        t = rs.integers(0, 1000, n)
        tgt = rs.standard_normal((n, 3)).astype(np.float32)
    out = O.resnet_fwd(g["params"], x, t, "f64")
    dout = (2.0 / (3 * n)) * (out - tgt)
    ref = O.resnet_bwd(g["params"], x, t, dout, "f64")
    dp = host(B.resnet_bwd(dev(g["params"]), dev(x), dev(t, torch.int64), dev(dout), 1000, precision=1))
    assert np.isfinite(dp).all()
    LS = 255 * 255 + 255
    for l in range(7):                                      # per layer: weights and biases, relative to the layer's gradient norm
        lo, hi = l * LS, (l + 1) * LS if l < 6 else dp.size
        err = np.linalg.norm(dp[lo:hi] - ref[lo:hi]) / np.linalg.norm(ref[lo:hi])
        assert err < 2.5e-2, (l, err)
    # the stash handed over from the training forward gives the same gradient bit for bit, and the same output
    o2, st = B.resnet_fwd_stash(dev(g["params"]), dev(x), dev(t, torch.int64), 1000, precision=1)
    assert torch.equal(o2, B.resnet_fwd(dev(g["params"]), dev(x), dev(t, torch.int64), 1000, precision=1))
    dp2 = B.resnet_bwd(dev(g["params"]), dev(x), dev(t, torch.int64), dev(dout), 1000, precision=1, stash=st)
    assert np.array_equal(host(dp2), dp)
    cos = float(dp @ ref / np.linalg.norm(dp) / np.linalg.norm(ref))
    assert cos > 0.9995, cos
    if n == 96:
        assert np.linalg.norm(dp - g["grad"]) / np.linalg.norm(g["grad"]) < 2.5e-2
    # fp32 path (exact fp32 MFMA, fp32 dumps): the reference's own precision
    dp32 = host(B.resnet_bwd(dev(g["params"]), dev(x), dev(t, torch.int64), dev(dout), 1000, precision=0))
    for l in range(7):
        lo, hi = l * LS, (l + 1) * LS if l < 6 else dp32.size
        err = np.linalg.norm(dp32[lo:hi] - ref[lo:hi]) / np.linalg.norm(ref[lo:hi])
        assert err < 2e-5, (l, err)
    assert np.abs(dp32 - ref).max() < 2e-5 * max(1.0, np.abs(ref).max())
    if n == 96:
        assert np.abs(dp32 - g["grad"]).max() < 2e-5 * max(1.0, np.abs(g["grad"]).max())


@pytest.mark.gpu
def test_wide_rotpredict_module_and_training_step(B, golden):
    """so3x.so3_lock_train.RotPredict: the reference's state_dict keys, forward == golden, autograd through the fused
    backward, SO3Diffusion dispatches its sampler to the wide-network chain kernel."""
    from so3x.so3_lock_train import RotPredict
    from so3x.diffusion import SO3Diffusion
    from so3x import rng
    g = golden["resnet"]
    net = RotPredict(out_type="skewvec", precision="bf16")
    assert list(net.state_dict().keys()) == [str(k) for k in g["param_names"]]
    off, sd = 0, {}
    for k, v in net.state_dict().items():
        sd[k] = torch.from_numpy(g["params"][off:off + v.numel()].reshape(v.shape).copy())
        off += v.numel()
    net.load_state_dict(sd)
    net = net.to(DEV)
    x, t = dev(g["x"]), dev(g["t"], torch.int64)
    with torch.no_grad():
        net.precision = "fp32"
        assert float((net(x, t) - dev(g["out"])).abs().max()) < 2e-5
        net.precision = "bf16"
    loss = torch.nn.functional.mse_loss(net(x, t), dev(g["target"]))
    assert abs(float(loss.detach()) - float(g["loss"])) < 2e-2 * float(g["loss"])
    loss.backward()
    got = torch.cat([p.grad.reshape(-1) for p in net.net.parameters()])
    ref = dev(g["grad"])
    assert float((got - ref).norm() / ref.norm()) < 2.5e-2
    net.precision = "fp32"                                   # the reference's precision end to end
    net.zero_grad()
    loss32 = torch.nn.functional.mse_loss(net(x, t), dev(g["target"]))
    assert abs(float(loss32.detach()) - float(g["loss"])) < 1e-5 * float(g["loss"])
    loss32.backward()
    got32 = torch.cat([p.grad.reshape(-1) for p in net.net.parameters()])
    assert float((got32 - ref).abs().max()) < 2e-5 * float(ref.abs().max())
    net.precision = "bf16"
    net.zero_grad()
    proc = SO3Diffusion(net, timesteps=30).to(DEV)
    rng.manual_seed(3)
    xs = proc.p_sample_loop((500,))
    assert xs.shape == (500, 3, 3) and not torch.isnan(xs).any()
    assert float((xs @ xs.transpose(-1, -2) - torch.eye(3, device=DEV)).abs().max()) < 1e-4
    l2 = proc(proc.p_sample_loop((64,)))                     # a p_losses training step end to end
    l2.backward()
    assert torch.isfinite(l2)


@pytest.mark.gpu
def test_resnet_chain_explicit_draws_and_edge_sizes(B, golden):
    """explicit (axes, unif) draws for one step reproduce the oracle's p_sample with the same draws; gradients at the
    workgroup-pass boundaries (n = 1, 257) and with one shared timestep"""
    g = golden["resnet"]
    T = 1000
    betas = O.cosine_beta_schedule(T)
    sched = O.schedule_from_betas(betas)
    sched_d = dev(B.schedule_from_betas(betas))
    trap_p = B.igso3_build_tables(sched_d[12])
    rs = np.random.default_rng(9)
    n, t = 130, 321
    x0 = O.quat_to_rmat(rs.standard_normal((n, 4)).astype(np.float32))
    axes = rs.standard_normal((n, 3)).astype(np.float32)
    unif = rs.random(n).astype(np.float32)
    out = host(B.resnet_p_sample_chain(dev(g["params"]), sched_d, trap_p, dev(x0), t, 1, axes=dev(axes), unif=dev(unif), precision=0))
    coef = [float(sched[i][t]) for i in (6, 7, 10, 11)]
    v = O.resnet_fwd(g["params"], x0, np.full(n, t), "f64")
    _, mean = O.p_mean(x0, v, *coef, "f64")
    smp, _ = O.igso3_sample(host(trap_p)[t:t + 1], axes, unif, prec="f64")
    ref = O.rmul(mean, smp, "f64")
    assert np.abs(out - ref).max() < 5e-5
    with pytest.raises(B.So3xError):          # explicit draws are for a single step
        B.resnet_p_sample_chain(dev(g["params"]), sched_d, trap_p, dev(x0), t, 2, axes=dev(axes), unif=dev(unif), precision=0)
    for m in (1, 257):
        xm = O.quat_to_rmat(rs.standard_normal((m, 4)).astype(np.float32))
        dout = rs.standard_normal((m, 3)).astype(np.float32) / m
        for tt in (rs.integers(0, T, m), np.array([77])):          # per-sample and (1,)-shaped t
            refg = O.resnet_bwd(g["params"], xm, tt, dout, "f64")
            got = host(B.resnet_bwd(dev(g["params"]), dev(xm), dev(tt, torch.int64), dev(dout), T, precision=0))
            assert np.abs(got - refg).max() < 3e-5 * max(1.0, np.abs(refg).max())


@pytest.mark.gpu
def test_wide_net_training_step_as_a_captured_graph(B):
    """the wide network's whole training step (stash forward, dX chain, dW GEMM, reduce, fused Adam) replays as one hipGraph"""
    from so3x.so3_lock_train import RotPredict
    from so3x.diffusion import SO3Diffusion
    from so3x.graphs import TrainStepGraph
    torch.manual_seed(0)
    net = RotPredict(out_type="skewvec", precision="bf16").to(DEV)
    proc = SO3Diffusion(net, timesteps=200).to(DEV)
    opt = torch.optim.Adam(net.parameters(), lr=1e-3, fused=True, capturable=True)
    x = B.quat_to_rmat(torch.randn(600, 4, device=DEV))
    before = torch.cat([p.detach().reshape(-1) for p in net.parameters()]).clone()
    g = TrainStepGraph(proc, opt, x.shape, warmup=2)
    # construction leaves no trace: the two warm-up steps (real updates on a placeholder batch) are undone -- parameters,
    # Adam's moments and step counts, the Philox counter
    assert torch.equal(torch.cat([p.detach().reshape(-1) for p in net.parameters()]), before) and int(proc.rng_counter) == 0
    assert all(float(st["step"]) == 0 and not st["exp_avg"].any() for st in opt.state.values())
    losses = [float(g.step(x)) for _ in range(5)]
    after = torch.cat([p.detach().reshape(-1) for p in net.parameters()])
    assert all(np.isfinite(losses)) and len(set(losses)) == 5
    assert float((after - before).abs().max()) > 1e-4 and torch.isfinite(after).all()
    assert int(proc.rng_counter) == 5                  # the replays only (capture records, it does not execute)
    assert all(float(st["step"]) == 5 for st in opt.state.values())


def _wide_from_flat(flat, precision):
    from so3x.so3_lock_train import RotPredict
    net = RotPredict(out_type="skewvec", precision=precision)
    sd, off = {}, 0
    for k, v in net.state_dict().items():
        sd[k] = torch.from_numpy(flat[off:off + v.numel()].reshape(v.shape).copy())
        off += v.numel()
    assert off == flat.size
    net.load_state_dict(sd)
    return net.to(DEV)


@pytest.mark.gpu
@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_G3_wide_net_chain_vs_reference_trained_population(B, golden, prec):
    """Gate G3 for the 255-wide network: weights the REFERENCE trained with its so3_lock_train recipe (the so3_lerp arc
    between two Euler rotations; tools/make_golden.py trained_chain_samples_wide), 2048 samples of this stack's 1000-step
    chain against 2048 samples of the reference's own p_sample_loop under the reference's kernel two-sample test."""
    import so3x
    from so3x.diffusion import SO3Diffusion
    g = golden["chain_samples_trained_wide"]
    ref = g["x_final"]
    m = len(ref)
    net = _wide_from_flat(g["params_f16"].astype(np.float32), prec)
    proc = SO3Diffusion(net, timesteps=1000).to(DEV)
    so3x.manual_seed(99)
    x = proc.p_sample_loop((m,))
    assert not torch.isnan(x).any()
    assert float((x @ x.transpose(-1, -2) - torch.eye(3, device=DEV)).abs().max()) < 1e-5
    mine = host(x)
    thr = O.ker_2samp_threshold(m)
    mmd = O.MMD(mine, ref)
    print(f"[wide {prec}] MMD {mmd:.2e} (bound {thr:.2e}; reference split halves {O.MMD(ref[:m // 2], ref[m // 2:]):.2e})")
    assert mmd < thr, (mmd, thr)
    assert mmd < 5 * max(O.MMD(ref[: m // 2], ref[m // 2:]), 1e-3)  # far tighter than the reference's bound: ~ estimator noise
    uni = O.quat_to_rmat(np.random.default_rng(0).standard_normal((m, 4)), "f64")
    assert O.MMD(uni, ref) > thr                      # the test has power against this population
    # the population lies along the training arc: same distance-to-arc statistics (arc = so3_lerp(R1, R2, w))
    w = np.linspace(0, 1, 201)
    arc = O.so3_lerp(np.repeat(g["R1"], 201, 0), np.repeat(g["R2"], 201, 0), w, "f64")

    def dist_to_arc(X):
        d = np.stack([O.rmat_dist(X.astype(np.float64), np.broadcast_to(a, X.shape).copy(), "f64") for a in arc[::10]], 0)
        return np.median(d.min(0))
    dm, dr = dist_to_arc(mine), dist_to_arc(ref)
    assert abs(dm - dr) < 0.3 * dr + 0.01, (dm, dr)


@pytest.mark.gpu
def test_wide_net_training_with_this_stack_reaches_the_reference_trained_population(B, golden):
    """the so3_lock_train recipe behind that fixture (1500 Adam steps of batch 128 at 3e-4 on the arc data) with THIS stack's
    kernels (bf16 operands), then the chain: statistically the same population as the reference's training + sampling"""
    import so3x
    from so3x.diffusion import SO3Diffusion
    from so3x.so3_lock_train import RotPredict
    from so3x import util
    g = golden["chain_samples_trained_wide"]
    ref = g["x_final"]
    m = len(ref)
    torch.manual_seed(0)
    so3x.manual_seed(5)
    net = RotPredict(out_type="skewvec", precision="bf16").to(DEV)
    proc = SO3Diffusion(net, timesteps=1000).to(DEV)
    opt = torch.optim.Adam(net.parameters(), lr=3e-4, fused=True)
    R1, R2 = dev(g["R1"]), dev(g["R2"])
    gen = torch.Generator(device=DEV).manual_seed(3)
    for i in range(1500):
        x0 = util.so3_lerp(R1, R2, torch.rand(128, 1, device=DEV, generator=gen))
        loss = proc(x0)
        opt.zero_grad()
        loss.backward()
        opt.step()
    assert float(loss.detach()) < 1.5
    x = proc.p_sample_loop((m,))
    mine = host(x)
    thr = O.ker_2samp_threshold(m)
    mmd = O.MMD(mine, ref)
    print(f"[wide trained here] MMD {mmd:.2e} (bound {thr:.2e})")
    assert not np.isnan(mine).any() and mmd < thr, (mmd, thr)


@pytest.mark.gpu
def test_wide_one_step_per_call_loop_runs_from_a_prepared_state(B):
    """so3_lock_test.py:24-31 drives the 255-wide network's sampler one `p_sample` per reverse step with a (1,)-shaped device t: as
    for the small network, every call is one launch from a cached preparation (so3x_resnet_p_sample_prepare / _prepared, t read on
    the device) -- bit-identical to one-step launches of the unprepared entry, prepared once, rebuilt when the parameters change."""
    from so3x import rng
    from so3x.diffusion import SO3Diffusion
    from so3x.so3_lock_train import RotPredict
    torch.manual_seed(0)
    net = RotPredict(out_type="skewvec", precision="bf16").to(DEV)
    T = 40
    proc = SO3Diffusion(net, timesteps=T).to(DEV)
    x0 = B.quat_to_rmat(torch.randn(1000, 4, device=DEV))
    calls = {"n": 0}
    real = B.resnet_p_sample_prepare

    def counting(*a, **k):
        calls["n"] += 1
        return real(*a, **k)
    B.resnet_p_sample_prepare = counting
    try:
        rng.manual_seed(4)
        x = x0
        for i in reversed(range(T)):
            x = proc.p_sample(x, torch.full((1,), i, device=DEV, dtype=torch.long))
        assert calls["n"] == 1
        rng.manual_seed(4)
        y = x0
        _, trap_p = proc._tables()
        prec = getattr(net, "chain_precision_code", net.precision_code)
        for k, i in enumerate(reversed(range(T))):
            y = B.resnet_p_sample_chain(net.flat_params_nograd(), proc._sched, trap_p, y, i, 1, seed=rng.seed(), rng_offset=k * T, precision=prec,
                                        guide_p=proc._guide_p)
        assert torch.equal(x, y) and torch.isfinite(x).all()
        rng.manual_seed(4)
        a = proc.p_sample(x0, 7)
        rng.manual_seed(4)
        b = proc.p_sample(x0, torch.full((1,), 7, device=DEV, dtype=torch.long))
        assert torch.equal(a, b) and calls["n"] == 1
        with torch.no_grad():
            net.net[0].layer[0].weight.mul_(1.5)          # an in-place torch update: the parameters' versions move
        rng.manual_seed(4)
        c = proc.p_sample(x0, 7)
        assert calls["n"] == 2 and not torch.equal(a, c)
    finally:
        B.resnet_p_sample_prepare = real
