"""The rotation-matrix head and the "prevstep" objective: RotPredict(out_type="rotmat")
(so3_train.py:19-22,47-48; so3_lock_train.py:19-22,57-58), six2rmat (util.py:67-76), autograd through log_rmat / rmat_dist
(util.py:164-192, 315-322) and SO3Diffusion(loss_type="prevstep") (diffusion.py:358-365).
CPU: the oracle's closed-form gradients against the reference's own torch-autograd results (tests/golden/prevstep.npz,
made by tools/make_golden.py prevstep).  GPU: the HIP kernels against the oracle and the fixture through the C ABI."""
import numpy as np
import pytest
import torch

from oracle import oracle as O

DEV = "cuda:0"
NETS = (("mlp", O.mlp_fwd, O.mlp_bwd), ("resnet", O.resnet_fwd, O.resnet_bwd))


def dev(a, dtype=torch.float32):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV).to(dtype)


def host(t):
    return t.detach().cpu().numpy()


# ----------------------------------------------------------------------------- CPU: oracle pinned by the reference
def test_oracle_six2rmat_and_its_gradient_vs_reference_autograd(golden):
    g = golden["prevstep"]
    for prec, tol in (("f32", 5e-6), ("f64", 5e-6)):       # the fixture is the reference's fp32 run
        assert np.abs(O.six2rmat(g["six_x"], prec) - g["six_R"]).max() < 2e-6
        assert np.abs(O.six2rmat_bwd(g["six_x"], g["six_G"], prec) - g["six_grad"]).max() < tol * np.abs(g["six_grad"]).max()
    R = O.six2rmat(g["six_x"], "f64")
    assert np.abs(R @ R.transpose(0, 2, 1) - np.eye(3)).max() < 1e-12 and np.abs(np.linalg.det(R) - 1).max() < 1e-12


def test_oracle_log_and_dist_gradients_vs_reference_autograd(golden):
    g = golden["prevstep"]
    for prec in ("f32", "f64"):
        dR = O.log_rmat_bwd(g["log_R"], g["log_G"], prec)
        # per-sample scale: the gradient grows like 1/(pi - omega) near pi (sample 63 of the fixture reaches 357)
        scale = np.maximum(1.0, np.abs(g["log_grad"]).reshape(64, -1).max(1))[:, None, None]
        assert (np.abs(dR - g["log_grad"]) / scale).max() < 2e-6
        da, db = O.rmat_dist_bwd(g["dist_a"], g["dist_b"], g["dist_g"], prec)
        # b = a exp(angle axis) with angles down to 1e-3: fp32 cancellation in a^T b leaves ~1e-5 absolute
        assert np.abs(da - g["dist_grad_a"]).max() < 2e-5 and np.abs(db - g["dist_grad_b"]).max() < 2e-5


@pytest.mark.parametrize("net,fwd,bwd", NETS)
def test_oracle_prevstep_training_step_vs_reference(golden, net, fwd, bwd):
    g = golden["prevstep"]
    pre = net + "_T100_"
    params = g[net + "_params"]
    assert params.size == (O.N_PARAMS_ROTMAT if net == "mlp" else O.N_PARAMS_RESNET_ROTMAT)
    sched = O.schedule_from_betas(O.cosine_beta_schedule(100))
    for prec in ("f32", "f64"):
        out6 = fwd(params, g[pre + "x_t"], g[pre + "t"], prec)
        assert out6.shape == (64, 6) and np.abs(out6 - g[pre + "out6"]).max() < 2e-6
        xr = O.six2rmat(out6, prec)
        assert np.abs(xr - g[pre + "x_recon"]).max() < 1e-5   # the normalisation divides the 1e-6 output error by |a| < 1
        step, d2, dxr = O.prevstep_loss(xr, g[pre + "x0"], g[pre + "x_t"], sched, g[pre + "t"], prec)
        assert np.abs(step - g[pre + "step"]).max() < 5e-6
        assert abs(d2.mean() - float(g[pre + "loss"])) < 2e-6 * float(g[pre + "loss"])
        d6 = O.six2rmat_bwd(out6, dxr / len(d2), prec)
        dp = bwd(params, g[pre + "x_t"], g[pre + "t"], d6, prec)
        assert np.abs(dp - g[pre + "grad"]).max() < 5e-6 * np.abs(g[pre + "grad"]).max()


def test_oracle_rotmat_head_leaves_the_skewvec_path_alone(golden):
    """a 6-row head is the 3-row head plus three more rows: same trunk, same first three outputs"""
    g = golden["prevstep"]
    p6 = g["mlp_params"]
    trunk = 4 * (65 * 65 + 65)
    p3 = np.concatenate([p6[:trunk + 3 * 65], p6[trunk + 6 * 65:trunk + 6 * 65 + 3]])
    x, t = g["mlp_T100_x_t"], g["mlp_T100_t"]
    assert np.array_equal(O.mlp_fwd(p3, x, t, "f32"), O.mlp_fwd(p6, x, t, "f32")[:, :3])


# ----------------------------------------------------------------------------- GPU
@pytest.fixture(scope="module")
def B():
    from so3x import backend
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return backend


@pytest.mark.gpu
def test_six2rmat_forward_backward_vs_reference(B, golden):
    g = golden["prevstep"]
    x = dev(g["six_x"]).requires_grad_(True)
    R = B.six2rmat(x)
    assert np.abs(host(R) - g["six_R"]).max() < 2e-6
    (R * dev(g["six_G"])).sum().backward()
    assert np.abs(host(x.grad) - g["six_grad"]).max() < 1e-5 * np.abs(g["six_grad"]).max()
    # ragged sizes, leading batch dims, zero-size
    for n in (1, 255, 257, 1000):
        xs = torch.randn(n, 6, device=DEV)
        err = np.abs(host(B.six2rmat(xs)) - O.six2rmat(host(xs), "f64"))
        assert err.max() < 5e-5 and np.median(err) < 2e-7   # nearly parallel a1, a2 amplify rounding by 1/|a2 - (b1.a2) b1|
    assert B.six2rmat(torch.randn(2, 5, 6, device=DEV)).shape == (2, 5, 3, 3)
    assert B.six2rmat(torch.empty(0, 6, device=DEV)).shape == (0, 3, 3)
    with pytest.raises(ValueError):
        B.six2rmat(torch.randn(4, 5, device=DEV))


@pytest.mark.gpu
def test_log_rmat_and_rmat_dist_are_differentiable_like_the_reference(B, golden):
    from so3x import util
    g = golden["prevstep"]
    R = dev(g["log_R"]).requires_grad_(True)
    L = util.log_rmat(R)
    assert np.abs(host(L) - g["log_out"]).max() < 2e-5
    (L * dev(g["log_G"])).sum().backward()
    scale = np.maximum(1.0, np.abs(g["log_grad"]).reshape(64, -1).max(1))[:, None, None]
    # near pi the gradient is ~1/(pi - omega)^2-conditioned: compare with the fp64 oracle at the oracle's own fp32 error
    ref64 = O.log_rmat_bwd(g["log_R"], g["log_G"], "f64")
    err32 = np.abs(O.log_rmat_bwd(g["log_R"], g["log_G"], "f32") - ref64) / scale
    assert (np.abs(host(R.grad) - ref64) / scale).max() < max(5e-6, 4 * err32.max())
    a = dev(g["dist_a"]).requires_grad_(True)
    b = dev(g["dist_b"]).requires_grad_(True)
    d = util.rmat_dist(a, b)
    assert np.abs(host(d) - g["dist"]).max() < 2e-5
    (d * dev(g["dist_g"])).sum().backward()
    assert np.abs(host(a.grad) - g["dist_grad_a"]).max() < 5e-5 and np.abs(host(b.grad) - g["dist_grad_b"]).max() < 5e-5
    # no grad requested: the plain kernels (no autograd node)
    assert not util.rmat_dist(a.detach(), b.detach()).requires_grad
    # the identity: distance 0, gradient 0 (the reference's norm backward gives 0 there as well)
    e = torch.eye(3, device=DEV).repeat(3, 1, 1).requires_grad_(True)
    util.rmat_dist(e, torch.eye(3, device=DEV).repeat(3, 1, 1)).sum().backward()
    assert torch.equal(e.grad, torch.zeros_like(e.grad))


@pytest.mark.gpu
@pytest.mark.parametrize("net", ["mlp", "resnet"])
@pytest.mark.parametrize("prec,tol_out,tol_grad", [("fp32", 5e-6, 2e-5), ("bf16", 3e-2, 6e-2)])
def test_rotmat_head_forward_backward_vs_oracle(B, golden, net, prec, tol_out, tol_grad):
    """the 6-wide head through the C ABI: raw outputs and parameter gradients for a random dL/dout [n, 6]"""
    g = golden["prevstep"]
    params = dev(g[net + "_params"])
    code = B.PREC_F32 if prec == "fp32" else B.PREC_BF16
    for n in (64, 300):
        rng = np.random.default_rng(n)
        q = rng.standard_normal((n, 4)).astype(np.float32)
        x = host(B.quat_to_rmat(dev(q)))
        t = rng.integers(0, 100, n)
        dout = rng.standard_normal((n, 6)).astype(np.float32)
        fwd_o, bwd_o = (O.mlp_fwd, O.mlp_bwd) if net == "mlp" else (O.resnet_fwd, O.resnet_bwd)
        ref = fwd_o(g[net + "_params"], x, t, "f64")
        gref = bwd_o(g[net + "_params"], x, t, dout, "f64")
        if net == "mlp":
            outs = [B.mlp_fwd(params, dev(x), dev(t, torch.int64), code, tt) for tt in (0, 100)]
            grads = [B.mlp_bwd(params, dev(x), dev(t, torch.int64), dev(dout), code, tt) for tt in (0, 100)]
            if prec == "bf16":
                o_s, zs = B.mlp_fwd_stash(params, dev(x), dev(t, torch.int64), 100)
                outs.append(o_s)
                grads.append(B.mlp_bwd(params, dev(x), dev(t, torch.int64), dev(dout), code, 100, zstash=zs))
        else:
            outs = [B.resnet_fwd(params, dev(x), dev(t, torch.int64), 100, code)]
            o_s, st = B.resnet_fwd_stash(params, dev(x), dev(t, torch.int64), 100, code)
            outs.append(o_s)
            grads = [B.resnet_bwd(params, dev(x), dev(t, torch.int64), dev(dout), 100, code),
                     B.resnet_bwd(params, dev(x), dev(t, torch.int64), dev(dout), 100, code, stash=st)]
        for o in outs:
            assert o.shape == (n, 6)
            assert np.abs(host(o) - ref).max() < tol_out * max(1.0, np.abs(ref).max())
        for gr in grads:
            assert gr.numel() == g[net + "_params"].size
            assert np.abs(host(gr) - gref).max() < tol_grad * np.abs(gref).max()
            head = host(gr)[-(6 * (65 if net == "mlp" else 255) + 6):]
            assert np.abs(head).min() > 0 or n < 8  # every head row received a gradient


@pytest.mark.gpu
@pytest.mark.parametrize("net", ["mlp", "resnet"])
def test_prevstep_training_step_vs_reference(B, golden, net):
    """SO3Diffusion(loss_type="prevstep") with the reference's weights and recorded draws: loss and every parameter gradient"""
    from so3x.diffusion import SO3Diffusion
    from so3x.so3_train import RotPredict
    from so3x.so3_lock_train import RotPredict as WideRotPredict
    g = golden["prevstep"]
    pre = net + "_T100_"
    model = (RotPredict if net == "mlp" else WideRotPredict)(out_type="rotmat", precision="fp32")
    names = [str(s) for s in g[net + "_param_names"]]
    assert list(model.state_dict().keys()) == names
    flat, off, sd = g[net + "_params"], 0, {}
    for k, v in model.state_dict().items():
        sd[k] = torch.from_numpy(flat[off:off + v.numel()].reshape(v.shape).copy())
        off += v.numel()
    model.load_state_dict(sd)
    model = model.to(DEV)
    proc = SO3Diffusion(model, timesteps=100, loss_type="prevstep").to(DEV)
    x0, t = dev(g[pre + "x0"]), dev(g[pre + "t"], torch.int64)
    loss = proc.p_losses(x0, t, axes=dev(g[pre + "axes"]), unif=dev(g[pre + "unif"]))
    assert abs(float(loss) - float(g[pre + "loss"])) < 1e-5 * float(g[pre + "loss"])
    loss.backward()
    grad = np.concatenate([host(p.grad).reshape(-1) for p in model.parameters()])
    assert np.abs(grad - g[pre + "grad"]).max() < 1e-4 * np.abs(g[pre + "grad"]).max()
    # the pieces: x_t, the step rotation, the network's rotation
    x_t = proc.q_sample(x0, t, axes=dev(g[pre + "axes"]), unif=dev(g[pre + "unif"]))
    assert np.abs(host(x_t) - g[pre + "x_t"]).max() < 1e-5
    assert np.abs(host(B.prevstep_step(proc._sched, x0, dev(g[pre + "x_t"]), t)) - g[pre + "step"]).max() < 1e-5
    with torch.no_grad():
        assert np.abs(host(model(dev(g[pre + "x_t"]), t)) - g[pre + "x_recon"]).max() < 1e-5
    # sampling with a rotation-matrix head is undefined in the reference (shape mismatch at diffusion.py:293): loud error
    with pytest.raises(ValueError, match="skewvec"):
        proc.p_sample(x_t, t)


@pytest.mark.gpu
def test_prevstep_loss_kernel_vs_oracle_ragged_and_large(B, golden):
    sched_np = B.schedule_from_betas(B.cosine_beta_schedule(1000))
    sched = dev(sched_np)
    for n in (1, 255, 4097, 1 << 18):
        rng = np.random.default_rng(n)
        mk = lambda: host(B.quat_to_rmat(dev(rng.standard_normal((n, 4)).astype(np.float32))))
        xr, xs, xn = mk(), mk(), mk()
        t = rng.integers(0, 1000, n)
        xr_d = dev(xr).requires_grad_(True)
        loss = B.prevstep_loss(sched, xr_d, dev(xs), dev(xn), dev(t, torch.int64))
        loss.backward()
        if n <= 4097:
            step, d2, dx = O.prevstep_loss(xr, xs, xn, sched_np, t, "f64")
            assert abs(float(loss) - d2.mean()) < 2e-5 * d2.mean()
            # d omega / dM is 1/(pi - omega)-conditioned: scale per sample
            sc = np.maximum(1.0, np.abs(dx).reshape(n, -1).max(1))[:, None, None]
            # ... and its fp32 error is 1/(pi - omega)^2-conditioned: the yardstick is the fp32 oracle's own error
            err32 = (np.abs(O.prevstep_loss(xr, xs, xn, sched_np, t, "f32")[2] - dx) / sc).max()
            assert (np.abs(host(xr_d.grad) * n - dx) / sc).max() < max(1e-4, 4 * err32)
            assert np.median(np.abs(host(xr_d.grad) * n - dx)) < 1e-5
        else:
            # size-independent property: the loss is a mean of per-sample terms -> equals the mean of the two halves' losses
            h = n // 2
            la = B.prevstep_loss(sched, dev(xr[:h]), dev(xs[:h]), dev(xn[:h]), dev(t[:h], torch.int64))
            lb = B.prevstep_loss(sched, dev(xr[h:]), dev(xs[h:]), dev(xn[h:]), dev(t[h:], torch.int64))
            assert abs(float(loss) - 0.5 * (float(la) + float(lb))) < 1e-5 * float(loss)
            # and x_recon == step has zero loss and zero gradient
            st = B.prevstep_step(sched, dev(xs), dev(xn), dev(t, torch.int64)).requires_grad_(True)
            l0 = B.prevstep_loss(sched, st, dev(xs), dev(xn), dev(t, torch.int64))
            l0.backward()
            assert float(l0) < 1e-9 and float(st.grad.abs().max()) < 1e-6


@pytest.mark.gpu
def test_prevstep_training_reduces_the_loss_and_runs_as_a_graph(B):
    """a few Adam steps on the prevstep objective with the bf16 kernels (eager), then the same step as a captured graph"""
    from so3x.diffusion import SO3Diffusion
    from so3x.so3_train import RotPredict
    from so3x.graphs import TrainStepGraph
    torch.manual_seed(0)
    net = RotPredict(out_type="rotmat", precision="bf16").to(DEV)
    proc = SO3Diffusion(net, timesteps=100, loss_type="prevstep").to(DEV)
    opt = torch.optim.Adam(net.parameters(), lr=2e-3, fused=True, capturable=True)
    x = B.quat_to_rmat(torch.randn(4096, 4, device=DEV))
    first = []
    for _ in range(5):
        opt.zero_grad(set_to_none=True)
        loss = proc(x)
        loss.backward()
        opt.step()
        first.append(float(loss))
    del loss  # a live loss pins the parameters' AccumulateGrad nodes to the eager stream, which torch cannot capture across
    g = TrainStepGraph(proc, opt, x.shape, warmup=2)
    later = [float(g.step(x)) for _ in range(150)]
    assert all(np.isfinite(first + later))
    assert np.mean(later[-20:]) < 0.8 * np.mean(first)


@pytest.mark.gpu
def test_fused_prevstep_with_six2rmat_inside_equals_the_composition(B):
    """so3x_prevstep_loss6 (six2rmat + loss + six2rmat backward in one kernel) against six2rmat -> so3x_prevstep_loss"""
    sched = dev(B.schedule_from_betas(B.cosine_beta_schedule(200)))
    for n in (1, 255, 3000):
        rng = np.random.default_rng(n)
        out6 = dev(rng.standard_normal((n, 6)).astype(np.float32))
        xs = B.quat_to_rmat(dev(rng.standard_normal((n, 4)).astype(np.float32)))
        xn = B.quat_to_rmat(dev(rng.standard_normal((n, 4)).astype(np.float32)))
        t = dev(rng.integers(0, 200, n), torch.int64)
        a = out6.clone().requires_grad_(True)
        la = B.prevstep_loss6(sched, a, xs, xn, t)
        la.backward()
        b = out6.clone().requires_grad_(True)
        lb = B.prevstep_loss(sched, B.six2rmat(b), xs, xn, t)
        lb.backward()
        assert abs(float(la) - float(lb)) <= 1e-6 * abs(float(lb))
        assert float((a.grad - b.grad).abs().max()) <= 1e-6 * max(1.0, float(b.grad.abs().max()))
