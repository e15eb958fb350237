"""GPU tests of the PyTorch-ROCm operator library (libso3x_torch.so, TORCH_LIBRARY(so3x, ...)): torch.library.opcheck on
the operators of the hot path (schema vs behaviour, fake-tensor kernels, autograd registration, AOT dispatch), the
dispatcher-level autograd of the score network, and agreement with the raw C ABI reached through ctypes."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def env():
    from so3x import backend as B
    from so3x.so3_train import RotPredict
    from so3x.diffusion import SO3Diffusion
    torch.manual_seed(0)
    net = RotPredict(out_type="skewvec", precision="bf16").to(DEV)
    proc = SO3Diffusion(net, timesteps=100).to(DEV)
    trap_q, trap_p = proc._tables()
    n = 200
    x = B.quat_to_rmat(torch.randn(n, 4, device=DEV))
    t = torch.randint(0, 100, (n,), device=DEV)
    return dict(B=B, ops=B.ops(), net=net, proc=proc, trap_q=trap_q, trap_p=trap_p, x=x, t=t, n=n)


def test_opcheck_on_the_hot_path_operators(env):
    from torch.library import opcheck
    ops, B, proc, x, t, n = env["ops"], env["B"], env["proc"], env["x"], env["t"], env["n"]
    params = env["net"].flat_data().clone()
    checks = ("test_schema", "test_faketensor", "test_autograd_registration", "test_aot_dispatch_dynamic")
    opcheck(ops.mlp_fwd.default, (params, x, t, 1, B.PREC_F32, 0), test_utils=checks)
    opcheck(ops.mlp_fwd.default, (params, x, t, 1, B.PREC_BF16, 100), test_utils=checks)
    opcheck(ops.mlp_fwd.default, (params.clone().requires_grad_(), x, t, 1, B.PREC_F32, 100), test_utils=checks)
    dout = torch.randn(n, 3, device=DEV)
    opcheck(ops.mlp_bwd.default, (params, x, t, 1, dout, B.PREC_F32, 100, None), test_utils=checks)
    opcheck(ops.q_sample_target.default, (proc._sched, env["trap_q"], proc._guide_q, x, t, True, None, None, None, 5, 0, None, 0, True,
                                           True, False), test_utils=checks)
    opcheck(ops.p_sample_chain.default, (params, proc._sched, env["trap_p"], proc._guide_p, x, 50, 3, None, None, 5, 0, 0,
                                         B.PREC_BF16), test_utils=checks)
    opcheck(ops.log_rmat_vec.default, (x,), test_utils=checks)
    opcheck(ops.so3_scale.default, (x, torch.rand(n, device=DEV), 1), test_utils=checks)
    opcheck(ops.igso3_logprob_score.default, (x, torch.rand(n, device=DEV) * 0.5 + 0.2, 1, True, False), test_utils=checks)
    m, v, step = torch.zeros_like(params), torch.zeros_like(params), torch.zeros(2, device=DEV)
    opcheck(ops.adam_step.default, (params.clone(), torch.randn_like(params), m, v, step, 1e-3, 0.9, 0.999, 1e-8, 0.0, 1.0),
            test_utils=("test_schema", "test_faketensor"))


def test_opcheck_on_the_training_step_operators(env):
    """train_fwd used to hand the caller's `t` back as an output (an alias the schema does not declare); now the
    timesteps the step ran with are always a fresh tensor.  Schema-vs-behaviour and fake kernels of the whole-step operators
    and of the stage operators a pipelined captured step is made of, with given and with in-kernel timesteps."""
    from torch.library import opcheck
    ops, B, proc, x, t, n = env["ops"], env["B"], env["proc"], env["x"], env["t"], env["n"]
    params = env["net"].flat_data().clone()
    checks = ("test_schema", "test_faketensor")
    T = proc.num_timesteps
    for tt, counter in ((t, None), (None, None), (None, torch.zeros(1, dtype=torch.int64, device=DEV))):
        opcheck(ops.train_fwd.default, (params, proc._sched, env["trap_q"], proc._guide_q, x, tt, True, None, None, 5, 0, counter, 0, True),
                test_utils=checks)
    loss, carry, _ = B.train_fwd(params, proc._sched, env["trap_q"], x, t, seed=5, guide_q=proc._guide_q)
    x_t, t_used, dout, zstash, ws = carry
    assert t_used.data_ptr() != t.data_ptr() and torch.equal(t_used, t)
    opcheck(ops.train_bwd.default, (x_t, t_used, dout, zstash, ws, T, None, params.numel()), test_utils=checks)
    opcheck(ops.train_bwd.default, (x_t, t_used, dout, zstash, ws, T, torch.full((1,), 0.5, device=DEV), params.numel()), test_utils=checks)
    buf = B.TrainBuffers(n, T, DEV, want_out=True)
    counter = torch.zeros(1, dtype=torch.int64, device=DEV)
    opcheck(ops.train_noise.default, (proc._sched, env["trap_q"], proc._guide_q, x, None, True, None, None, 5, 0, counter, 0, buf.x_t, buf.t_used,
                                      buf.workspace), test_utils=checks)
    opcheck(ops.train_noise.default, (proc._sched, env["trap_q"], proc._guide_q, x, t, True, None, None, 5, 0, None, 0, buf.x_t, buf.t_used,
                                      buf.workspace), test_utils=checks)
    # (opcheck runs an operator on COPIES of its arguments: the buffers themselves are still torch.empty here; the kernels clamp
    #  whatever timesteps they are handed -- an uninitialised t_used once indexed the bias table out of its aperture)
    opcheck(ops.train_net.default, (params, T, buf.x_t, buf.t_used, buf.dout, buf.zstash, buf.loss, buf.out, counter, buf.workspace),
            test_utils=checks)
    B.train_noise(buf, proc._sched, env["trap_q"], x, t, seed=5, guide_q=proc._guide_q)
    B.train_net(buf, params)
    opcheck(ops.train_bwd_partial.default, (buf.x_t, buf.t_used, buf.dout, buf.zstash, T, buf.workspace), test_utils=checks)
    opcheck(ops.train_bwd_reduce.default, (n, T, None, buf.grad, buf.workspace), test_utils=checks)
    for tt, ctr in ((t, None), (None, counter)):   # the one-kernel step: every optional output present / absent
        opcheck(ops.train_fused.default, (params, proc._sched, env["trap_q"], proc._guide_q, x, tt, True, None, None, 5, 0, ctr, 0, buf.loss,
                                          buf.t_used, buf.x_t, buf.out, buf.workspace), test_utils=checks)
        opcheck(ops.train_fused.default, (params, proc._sched, env["trap_q"], proc._guide_q, x, tt, True, None, None, 5, 0, ctr, 0, buf.loss,
                                          None, None, None, buf.workspace), test_utils=checks)
    m_, v_, st_ = torch.zeros_like(params), torch.zeros_like(params), torch.zeros(2, device=DEV)
    opcheck(ops.train_bwd_reduce_adam.default, (n, T, None, buf.grad, buf.workspace, params.clone(), m_, v_, st_, 1e-3, 0.9, 0.999, 1e-8, 0.0, 1.0),
            test_utils=checks)
    # reduction + Adam in one launch == the two launches, bit for bit (parameters, moments, step count, gradient)
    B.train_noise(buf, proc._sched, env["trap_q"], x, t, seed=5, guide_q=proc._guide_q)
    B.train_net(buf, params)
    B.train_bwd_partial(buf)
    pa, ma, va, sa = params.clone(), torch.zeros_like(params), torch.zeros_like(params), torch.zeros(2, device=DEV)
    pb, mb, vb, sb = params.clone(), torch.zeros_like(params), torch.zeros_like(params), torch.zeros(2, device=DEV)
    for _ in range(3):
        ga = B.train_bwd_reduce(buf).clone()
        B.adam_step(pa, ga, ma, va, sa, 1e-3, 0.9, 0.999, 1e-8, 0.0, 0.5)
        gb = B.train_bwd_reduce_adam(buf, pb, mb, vb, sb, 1e-3, 0.9, 0.999, 1e-8, 0.0, 0.5).clone()
        assert torch.equal(ga, gb) and torch.equal(pa, pb) and torch.equal(ma, mb) and torch.equal(va, vb) and torch.equal(sa, sb)
    assert float(sa[0]) == 3.0 and not torch.equal(pa, params)
    # the stages compose to the whole-step operators, bit for bit
    B.train_noise(buf, proc._sched, env["trap_q"], x, t, seed=5, guide_q=proc._guide_q)
    B.train_net(buf, params)
    B.train_bwd_partial(buf)
    g = B.train_bwd_reduce(buf)
    assert torch.equal(buf.loss[0], loss) and torch.equal(buf.x_t, x_t) and torch.equal(buf.dout, dout)
    assert torch.equal(g, B.train_bwd(carry, params.numel(), T))


def test_opcheck_on_the_widened_rows(env):
    """the operators of SURVEY 8f's rows (SE(3) layer, statistics, the 255-wide network, the rotation-matrix head and its
    objective): schema, fake kernels, autograd registration, AOT dispatch -- and six2rmat's registered backward against a
    finite difference"""
    from torch.library import opcheck
    ops, B, proc, x, t, n = env["ops"], env["B"], env["proc"], env["x"], env["t"], env["n"]
    checks = ("test_schema", "test_faketensor", "test_autograd_registration", "test_aot_dispatch_dynamic")
    sh = torch.randn(n, 3, device=DEV)
    opcheck(ops.se3_q_sample_target.default, (proc._sched, env["trap_q"], proc._guide_q, 10.0, x, sh, t, True, None, None, None, 3, 0, 0, True),
            test_utils=checks)
    opcheck(ops.se3_p_mean.default, (proc._sched, x, sh, torch.randn(n, 3, device=DEV), torch.randn(n, 3, device=DEV), 40), test_utils=checks)
    opcheck(ops.se3_p_noise.default, (env["trap_p"][40].contiguous(), 0.3, 10.0, x, sh, None, None, None, 3, 0, 0, True), test_utils=checks)
    pos = torch.randn(n, 7, 3, device=DEV)
    opcheck(ops.rigid_move.default, (x, sh, pos, None), test_utils=checks)
    opcheck(ops.rigid_move.default, (x, sh, pos, B.quat_to_rmat(torch.randn(n, 7, 4, device=DEV))), test_utils=checks)
    opcheck(ops.rotate_cloud.default, (x, torch.randn(11, 3, device=DEV), 0, 11), test_utils=checks)
    opcheck(ops.kernel_sum.default, (x, x[:50].contiguous(), B.KERNEL_GAUSSIAN, 1.0), test_utils=checks)
    opcheck(ops.mse_loss.default, (sh, torch.randn(n, 3, device=DEV)), test_utils=checks)
    opcheck(ops.mse_grad.default, (sh, torch.randn(n, 3, device=DEV), torch.ones(1, device=DEV)), test_utils=checks)
    x6 = torch.randn(n, 6, device=DEV)
    opcheck(ops.six2rmat.default, (x6,), test_utils=checks)
    opcheck(ops.six2rmat.default, (x6.clone().requires_grad_(),), test_utils=checks)
    opcheck(ops.log_rmat_bwd.default, (x, torch.randn(n, 3, 3, device=DEV)), test_utils=checks)
    opcheck(ops.rmat_dist_bwd.default, (x, x.flip(0).contiguous(), torch.randn(n, device=DEV)), test_utils=checks)
    opcheck(ops.prevstep_loss.default, (proc._sched, x, x.flip(0).contiguous(), x.roll(1, 0).contiguous(), t, 1, True, False), test_utils=checks)
    opcheck(ops.prevstep_loss6.default, (proc._sched, x6, x, x.roll(1, 0).contiguous(), t, 1), test_utils=checks)
    from so3x.so3_lock_train import RotPredict as WideNet
    wide = WideNet(out_type="skewvec", precision="bf16").to(DEV)
    wp = wide.flat_data().clone()
    opcheck(ops.resnet_fwd.default, (wp, x, t, 1, 3, B.PREC_BF16, 100), test_utils=checks)
    opcheck(ops.resnet_p_sample_chain.default, (wp, proc._sched, env["trap_p"], proc._guide_p, x, 50, 2, None, None, 5, 0, 0, B.PREC_BF16),
            test_utils=checks)
    # the registered backward of six2rmat: directional finite difference in fp32
    a = x6[:16].clone().requires_grad_()
    w = torch.randn(16, 3, 3, device=DEV)
    (ops.six2rmat(a) * w).sum().backward()
    d = torch.randn_like(a)
    h = 1e-3
    fd = ((ops.six2rmat(a.detach() + h * d) - ops.six2rmat(a.detach() - h * d)) * w).sum() / (2 * h)
    assert abs(float((a.grad * d).sum()) - float(fd)) < 2e-3 * max(1.0, abs(float(fd)))


def test_dispatcher_level_autograd_of_the_score_network(env, golden):
    """torch.ops.so3x.mlp_fwd is differentiable in its parameters through the registered formula (so3x_mlp_bwd):
    the reference's autograd gradients of the seed-0 network (tests/golden/score_mlp.npz), fp32"""
    ops, B = env["ops"], env["B"]
    g = golden["score_mlp"]
    params = torch.from_numpy(np.concatenate([g[f"net_{l}_{k}"].reshape(-1) for l in (0, 2, 4, 6, 8) for k in ("weight", "bias")])).to(DEV)
    params.requires_grad_()
    x, t = torch.from_numpy(g["x"]).to(DEV), torch.from_numpy(g["t"]).to(DEV)
    out = ops.mlp_fwd(params, x, t, 1, B.PREC_F32, 0)
    assert out.requires_grad
    assert np.abs(out.detach().cpu().numpy() - g["out"]).max() < 2e-5 * np.abs(g["out"]).max()
    loss = ((out - torch.from_numpy(g["target"]).to(DEV)) ** 2).mean()           # F.mse_loss, as the fixture's loss
    assert abs(float(loss.detach()) - float(g["loss"])) < 2e-5 * float(g["loss"])
    loss.backward()
    ref = np.concatenate([g[f"grad_net_{l}_{k}"].reshape(-1) for l in (0, 2, 4, 6, 8) for k in ("weight", "bias")])
    assert np.abs(params.grad.cpu().numpy() - ref).max() < 1e-4 * np.abs(ref).max()


def test_operators_equal_the_raw_c_abi(env):
    """the operator library adds nothing to the numbers: an op and a direct ctypes call of the C entry point it wraps
    (the maintainer stub of INTEGRATION.md) give bit-identical results"""
    B, ops, x, n = env["B"], env["ops"], env["x"], env["n"]
    lib = B.lib()
    ref = torch.empty(n, 3, device=DEV)
    rc = lib.so3x_log_rmat_vec(C.c_void_p(torch.cuda.current_stream().cuda_stream), C.c_void_p(x.data_ptr()), C.c_void_p(ref.data_ptr()),
                               C.c_int64(n))
    assert rc == 0 and torch.equal(ops.log_rmat_vec(x), ref)
    k = torch.rand(n, device=DEV)
    ref9 = torch.empty_like(x)
    rc = lib.so3x_so3_scale(C.c_void_p(torch.cuda.current_stream().cuda_stream), C.c_void_p(x.data_ptr()), C.c_void_p(k.data_ptr()),
                            C.c_int64(1), C.c_void_p(ref9.data_ptr()), C.c_int64(n))
    assert rc == 0 and torch.equal(ops.so3_scale(x, k, 1), ref9)


def test_errors_surface_as_so3x_errors(env):
    B, x = env["B"], env["x"]
    with pytest.raises(B.So3xError, match="no CPU path"):
        B.log_rmat(x.cpu())
    with pytest.raises(B.So3xError, match="p_sample_chain failed"):
        B.p_sample_chain(env["net"].flat_data(), env["proc"]._sched, env["trap_p"], x, 3, 10)     # t_start - n_steps + 1 < 0
