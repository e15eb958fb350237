"""GPU tests of the training step (BASELINE config 4's per-GPU work): the three-call C ABI so3x_train_fwd /
so3x_train_bwd / so3x_adam_step behind SO3Diffusion.forward + loss.backward() + so3x.optim.Adam, against the reference's
own training-step goldens, the CPU oracle, the composed (kernel-per-op) path, and itself as a captured hipGraph --
single-process and as two data-parallel ranks sharing the one GPU of the test box."""
import copy
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import oracle as O

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def dev(a, dtype=torch.float32):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV).to(dtype)


def host(t):
    return t.detach().cpu().numpy()


@pytest.fixture(scope="module")
def mods():
    from so3x import util, diffusion, so3_train, backend, optim, rng
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return dict(util=util, diff=diffusion, train=so3_train, B=backend, optim=optim, rng=rng)


def _golden_net(mods, golden, precision):
    g = golden["score_mlp"]
    n = mods["train"].RotPredict(out_type="skewvec", precision=precision)
    n.load_state_dict({f"net.{l}.{k}": torch.from_numpy(g[f"net_{l}_{k}"]) for l in (0, 2, 4, 6, 8) for k in ("weight", "bias")})
    return n.to(DEV)


def test_fused_step_reproduces_the_reference_training_step(mods, golden):
    """p_losses -> backward through the fast path (bf16 operands) on the reference's recorded draws: loss and all 17,358
    gradients of the reference's own training step, to bf16-operand accuracy; and equal to the composed path
    (q_sample_target, mlp_fwd_stash, mse_loss, mlp_bwd as separate ops) to fp32 rounding."""
    B = mods["B"]
    g = golden["train_step"]
    net = _golden_net(mods, golden, "bf16")
    for T, seed, kernel in [(T, seed, k) for T in (100, 1000) for seed in (0, 1, 2) for k in ("fused", "staged")]:
        if True:
            pre = f"T{T}_s{seed}_"
            proc = mods["diff"].SO3Diffusion(net, timesteps=T, betas=golden["schedule"][f"betas64_{T}"]).to(DEV)
            proc.train_step_kernel = kernel   # the one-kernel step (round 4, the default) and round 3's three launches
            x0, t = dev(g[pre + "x0"]), dev(g[pre + "t"], torch.int64)
            ax, un = dev(g[pre + "axes"]), dev(g[pre + "unif"])
            net.zero_grad(set_to_none=True)
            loss = proc.p_losses(x0, t, axes=ax, unif=un)
            assert loss.grad_fn is not None and "FusedSkewvecLoss" in type(loss.grad_fn).__name__
            loss.backward()
            flat = net.flat_grad()
            assert flat is not None and flat.numel() == 17358                     # .grad = views of ONE flat tensor
            ref = g[pre + "grad_flat"]
            # bf16 operands through the forward AND the backward (the fp32 path holds 1e-4: test_gpu_parity.py)
            assert abs(float(loss.detach()) - float(g[pre + "loss"])) < 1e-2 * float(g[pre + "loss"])
            assert np.abs(host(flat) - ref).max() < 5e-2 * np.abs(ref).max()
            assert np.linalg.norm(host(flat) - ref) < 4e-2 * np.linalg.norm(ref)
            # composed path, same operands
            trap_q, _ = proc._tables()
            x_t, target, _ = B.q_sample_target(proc._sched, trap_q, x0, t, quirk_col0=True, axes=ax, unif=un, guide_q=proc._guide_q)
            params = net.flat_data()
            out, zs = B.mlp_fwd_stash(params, x_t, t, T)
            n = x0.shape[0]
            l2 = ((out - target) ** 2).mean()
            dout = (out - target) * (2.0 / (3 * n))
            g2 = B.mlp_bwd(params, x_t, t, dout, B.PREC_BF16, T, zstash=zs)
            # staged: the same kernels -> fp32 rounding; one-kernel: the same forward (loss to rounding), the backward multiplies by
            # the silu' the forward parked (table, f16) instead of re-evaluating it from f16 pre-activations
            assert abs(float(loss.detach()) - float(l2)) < (2e-6 if kernel == "staged" else 2e-5) * float(l2)
            assert float((flat - g2).abs().max()) < (2e-6 if kernel == "staged" else 1e-2) * float(g2.abs().max())


def test_fused_step_pieces_vs_oracle(mods, golden):
    """train_fwd's outputs one by one against the f64 oracle on the same draws: x_t (G1), the network output, the loss,
    d loss / d out"""
    B = mods["B"]
    g = golden["train_step"]
    net = _golden_net(mods, golden, "bf16")
    pre = "T1000_s1_"
    T = 1000
    proc = mods["diff"].SO3Diffusion(net, timesteps=T, betas=golden["schedule"]["betas64_1000"]).to(DEV)
    x0, t = dev(g[pre + "x0"]), dev(g[pre + "t"], torch.int64)
    trap_q, _ = proc._tables()
    loss, carry, out = B.train_fwd(net.flat_data(), proc._sched, trap_q, x0, t, quirk_col0=True, axes=dev(g[pre + "axes"]),
                                   unif=dev(g[pre + "unif"]), guide_q=proc._guide_q, want_out=True)
    x_t, tt, dout, _, _ = carry
    assert np.abs(host(x_t) - g[pre + "x_t"]).max() < 1e-5
    params = host(net.flat_data())
    ref_out = O.mlp_fwd(params, g[pre + "x_t"], g[pre + "t"], "f64")
    assert np.abs(host(out) - ref_out).max() < 2e-2 * max(1.0, np.abs(ref_out).max())
    tgt = g[pre + "target"]
    n = x0.shape[0]
    assert abs(float(loss) - float(((host(out).astype(np.float64) - tgt) ** 2).mean())) < 1e-6 * float(loss)
    assert np.abs(host(dout) - (host(out) - tgt) * (2.0 / (3 * n))).max() < 1e-7 * max(1.0, np.abs(tgt).max())


@pytest.mark.parametrize("quirk", [True, False])
def test_fused_step_at_the_shard_size_of_config_4(mods, quirk):
    """2^19 samples, BASELINE config 4's per-GPU shard, through so3x_train_fwd / so3x_train_bwd (k_mlp_fwd_stash +
    k_bwd_fused<stashed>): finite; deterministic; additive over a split of the batch (loss and gradient are sample means,
    the noise and the in-kernel timesteps are keyed by the global sample index); equal to the generic staged backward
    (k_bwd_stage + k_bwd_dw, the forward recomputed, per-sample sin/cos) on the same x_t and d loss / d out."""
    B = mods["B"]
    torch.manual_seed(0)
    net = mods["train"].RotPredict(out_type="skewvec", precision="bf16").to(DEV)
    T = 1000
    # quirk=True is SO3Diffusion's default (distributions.py:42-43: column 0 = sample 0's row): with drawn timesteps that row
    # is GLOBAL sample 0's whatever the shard, so the split below must not change a bit either way
    proc = mods["diff"].SO3Diffusion(net, timesteps=T, quirk_col0=quirk).to(DEV)
    n = 1 << 19
    x0 = B.quat_to_rmat(torch.randn(n, 4, device=DEV, generator=torch.Generator(device=DEV).manual_seed(3)))
    trap_q, _ = proc._tables()
    params = net.flat_data()

    def step(lo, hi):
        loss, carry, _ = B.train_fwd(params, proc._sched, trap_q, x0[lo:hi], None, quirk_col0=quirk, seed=5, rng_offset=17,
                                     index_base=lo, guide_q=proc._guide_q)
        return loss, carry, B.train_bwd(carry, 17358, T)

    loss, carry, grad = step(0, n)
    x_t, t, dout, _, _ = carry
    assert torch.isfinite(loss) and torch.isfinite(grad).all() and torch.isfinite(x_t).all()
    assert int(t.min()) == 0 and int(t.max()) == T - 1 and abs(float(t.double().mean()) - (T - 1) / 2) < 2.0   # uniform on {0..T-1}
    loss_b, _, grad_b = step(0, n)
    assert torch.equal(loss, loss_b) and torch.equal(grad, grad_b)                                              # deterministic
    cut = 200_000
    la, ca, ga = step(0, cut)
    lb, cb, gb = step(cut, n)
    assert torch.equal(ca[1], t[:cut]) and torch.equal(cb[1], t[cut:])          # the drawn timesteps do not depend on the split
    assert torch.equal(ca[0], x_t[:cut]) and torch.equal(cb[0], x_t[cut:])      # nor does the noise
    wa, wb = cut / n, (n - cut) / n
    assert abs(float(loss) - (wa * float(la) + wb * float(lb))) < 1e-5 * float(loss)
    assert float((grad - (wa * ga + wb * gb)).abs().max()) < 2e-4 * float(grad.abs().max())
    staged = B.mlp_bwd(params, x_t, t, dout, B.PREC_BF16, t_table=0)
    assert float((grad - staged).abs().max()) < 2e-3 * float(staged.abs().max())


def test_adam_kernel_vs_torch_golden(mods, golden):
    """so3x_adam_step on the gradients torch.optim.Adam was given (tests/golden/adam.npz): parameters after each of 25 steps"""
    B = mods["B"]
    g = golden["adam"]
    p = dev(g["p0"]).clone()
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    step = torch.zeros(2, device=DEV)
    for i, (grad, want) in enumerate(zip(g["grads"], g["params"])):
        B.adam_step(p, dev(grad), m, v, step, float(g["lr"]), 0.9, 0.999, 1e-8)
        assert np.abs(host(p) - want).max() <= 2e-7 * max(1.0, float(np.abs(want).max())), i
    assert float(step[0]) == 25.0 and float(step[1]) == 0.0                    # count advanced on the device, ticket back to 0
    # grad_scale: the 1/world of a summed all-reduce inside the update
    p2, m2, v2, s2 = dev(g["p0"]).clone(), torch.zeros_like(p), torch.zeros_like(p), torch.zeros(2, device=DEV)
    B.adam_step(p2, dev(g["grads"][0]) * 8, m2, v2, s2, float(g["lr"]), 0.9, 0.999, 1e-8, grad_scale=0.125)
    assert np.abs(host(p2) - g["params"][0]).max() <= 2e-7


def test_so3x_adam_follows_torch_adam_on_the_network(mods):
    """so3x.optim.Adam (one launch on the flat buffers) against torch.optim.Adam on a twin network fed the same gradients"""
    torch.manual_seed(1)
    a = mods["train"].RotPredict(out_type="skewvec").to(DEV)
    b = copy.deepcopy(a)
    oa = mods["optim"].Adam(a, lr=3e-4)
    ob = torch.optim.Adam(b.parameters(), lr=3e-4)
    gen = torch.Generator(device=DEV).manual_seed(2)
    for _ in range(10):
        gflat = torch.randn(17358, device=DEV, generator=gen)
        a._install_flat_grad(gflat.clone())
        off = 0
        for p in b.parameters():
            p.grad = gflat[off:off + p.numel()].view_as(p).clone()
            off += p.numel()
        oa.step()
        ob.step()
    assert oa.step_count == 10
    pa, pb = a.flat_data(), torch.cat([p.detach().reshape(-1) for p in b.parameters()])
    assert float((pa - pb).abs().max()) < 5e-7


def test_frozen_parameters_stay_frozen(mods):
    """the flat-buffer optimizer runs ONE launch over all 17,358 values and the fused backward returns a full flat
    gradient; a parameter with requires_grad False must nevertheless keep its value and moments, as under torch.optim.Adam
    (which skips parameters whose .grad is None) -- twin networks, one layer frozen, same gradients"""
    torch.manual_seed(2)
    a = mods["train"].RotPredict(out_type="skewvec", precision="bf16").to(DEV)
    b = copy.deepcopy(a)
    for net in (a, b):
        net.net[2].weight.requires_grad_(False)
        net.net[2].bias.requires_grad_(False)
    proc = mods["diff"].SO3Diffusion(a, timesteps=100).to(DEV)
    oa = mods["optim"].Adam(a, lr=1e-3, weight_decay=1e-2)
    ob = torch.optim.Adam(b.parameters(), lr=1e-3, weight_decay=1e-2)
    x = mods["util"].quat_to_rmat(torch.randn(1024, 4, device=DEV))
    w0, b0 = a.net[2].weight.detach().clone(), a.net[2].bias.detach().clone()
    for _ in range(5):
        oa.zero_grad(set_to_none=True)
        loss = proc(x)                      # the one-kernel step: a full flat gradient comes back
        loss.backward()
        assert a.net[2].weight.grad is None and a.net[0].weight.grad is not None
        for pa, pb in zip(a.parameters(), b.parameters()):
            pb.grad = None if pa.grad is None else pa.grad.detach().clone()
        oa.step()
        ob.step()
    assert torch.equal(a.net[2].weight, w0) and torch.equal(a.net[2].bias, b0)
    for pa, pb in zip(a.parameters(), b.parameters()):
        assert float((pa - pb).abs().max()) < 5e-7
    from so3x.graphs import TrainStepGraph
    g = TrainStepGraph(proc, oa, x.shape)   # not every parameter trains: the generic serial form, whose optimizer step keeps them
    assert not g.fused
    g.step(x)
    assert torch.equal(a.net[2].weight, w0)
    with pytest.raises(ValueError, match="frozen"):
        oa.step_with_reduction(None)


def _run_steps(mods, base, x, mode, steps=4, optimizer="so3x", kernel="fused"):
    from so3x.graphs import TrainStepGraph
    net = copy.deepcopy(base)
    proc = mods["diff"].SO3Diffusion(net, timesteps=100).to(DEV)
    proc.train_step_kernel = kernel
    proc.rng_counter = torch.zeros(1, dtype=torch.int64, device=DEV)
    mods["rng"].manual_seed(7)
    make_opt = (lambda: mods["optim"].Adam(net, lr=1e-3)) if optimizer == "so3x" else \
        (lambda: torch.optim.Adam(net.parameters(), lr=1e-3, fused=True, capturable=True))
    opt = make_opt()
    if mode != "eager":
        # NO rewind here: construction (warm-up steps on a placeholder batch + capture) must leave parameters, optimizer state and
        # counters exactly as it found them (the first replay is the first eager step)
        g = TrainStepGraph(proc, opt, x.shape, warmup=2, pipeline={"graph": "auto", "serial": False, "pipelined": True, "staged": "staged", "fused": "fused"}[mode])
        assert g.pipelined == (mode == "pipelined")   # round 3's pipelined stages: on request only
        assert g.staged == (mode == "staged")         # round 3's stages as one stream, reduction + Adam as ONE launch
        # "auto" = the one-kernel step + [reduction + Adam] wherever the path allows (so3x.optim.Adam, train_step_kernel "fused")
        assert g.fused == (kernel == "fused" and optimizer == "so3x" and mode in ("graph", "fused"))
        assert torch.equal(net.flat_data(), base.flat_data()) and int(proc.rng_counter) == 0
    losses = []
    for _ in range(steps):
        if mode == "eager":
            opt.zero_grad(set_to_none=True)
            loss = proc(x)
            loss.backward()
            opt.step()
            losses.append(float(loss.detach()))
        else:
            losses.append(float(g.step(x)))
    if mode != "eager":
        g.flush()   # pipelined form: the last step's reduction + update (a no-op for the serial form)
    return losses, net.flat_data().clone(), int(proc.rng_counter), net, proc


def test_graph_replay_equals_the_eager_loop_bit_for_bit(mods):
    """so3x.graphs.TrainStepGraph: noising (fresh Philox offset from the device counter, timesteps drawn in the kernel),
    network, loss, backward and Adam (step count on the device) replayed as ONE hipGraph give exactly the losses and
    parameters of the same steps run eagerly -- every per-step quantity lives on the device"""
    torch.manual_seed(0)
    base = mods["train"].RotPredict(out_type="skewvec", precision="bf16").to(DEV)
    x = mods["util"].quat_to_rmat(torch.randn(2048, 4, device=DEV))
    for optimizer, kernel, modes in (("so3x", "fused", ("serial", "graph", "fused")), ("so3x", "staged", ("pipelined", "serial", "graph", "staged")),
                                     ("torch", "fused", ("graph",)), ("torch", "staged", ("graph",))):
        le, pe, ce, _, _ = _run_steps(mods, base, x, "eager", optimizer=optimizer, kernel=kernel)
        for mode in modes:
            lg, pg, cg, _, _ = _run_steps(mods, base, x, mode, optimizer=optimizer, kernel=kernel)
            assert ce == cg == 4
            assert len(set(lg)) == 4 and all(np.isfinite(lg))              # different noise and timesteps on every replay
            assert le == lg, (optimizer, kernel, mode, le, lg)
            assert torch.equal(pe, pg), (optimizer, kernel, mode)
            assert not torch.equal(pe, base.flat_data())


def test_pipelined_step_is_one_update_behind_until_flushed(mods):
    """the pipelined graph's contract (so3x/graphs.py): after step k the loss is batch k's and the parameters carry the updates
    up to k-1; flush() applies the outstanding one; stepping on after a flush continues the same trajectory; different batches per
    step are consumed by the step they are passed to"""
    from so3x.graphs import TrainStepGraph
    torch.manual_seed(0)
    base = mods["train"].RotPredict(out_type="skewvec", precision="bf16").to(DEV)
    xs = [mods["util"].quat_to_rmat(torch.randn(1536, 4, device=DEV, generator=torch.Generator(device=DEV).manual_seed(i))) for i in range(5)]

    def fresh():
        net = copy.deepcopy(base)
        proc = mods["diff"].SO3Diffusion(net, timesteps=100).to(DEV)
        proc.train_step_kernel = "staged"   # the pipelined form is made of round 3's stages
        proc.rng_counter = torch.zeros(1, dtype=torch.int64, device=DEV)
        mods["rng"].manual_seed(11)
        return net, proc, mods["optim"].Adam(net, lr=1e-3)

    net, proc, opt = fresh()
    eager_params, eager_losses = [], []
    for x in xs:
        opt.zero_grad(set_to_none=True)
        loss = proc(x)
        loss.backward()
        opt.step()
        eager_losses.append(float(loss.detach()))
        eager_params.append(net.flat_data().clone())
    del loss
    net, proc, opt = fresh()
    g = TrainStepGraph(proc, opt, xs[0].shape, pipeline=True)
    assert g.pipelined and g.mode == "in_graph"
    for k, x in enumerate(xs[:3]):
        assert float(g.step(x)) == eager_losses[k]
        want = base.flat_data() if k == 0 else eager_params[k - 1]
        assert torch.equal(net.flat_data(), want), k                      # one update behind
    g.flush()
    assert torch.equal(net.flat_data(), eager_params[2])
    g.flush()                                                             # idempotent
    assert torch.equal(net.flat_data(), eager_params[2]) and opt.step_count == 3
    for k in (3, 4):                                                      # the pipeline starts again behind a flush
        assert float(g.step(xs[k])) == eager_losses[k]
    g.flush()
    assert torch.equal(net.flat_data(), eager_params[4]) and opt.step_count == 5 and int(proc.rng_counter) == 5


def test_graph_refuses_changed_hyper_parameters(mods):
    """lr / betas / eps / weight_decay are kernel arguments frozen into the captured launches: a replay after an
    edit of param_groups must raise instead of silently training with the old value"""
    from so3x.graphs import TrainStepGraph
    torch.manual_seed(0)
    net = mods["train"].RotPredict(out_type="skewvec", precision="bf16").to(DEV)
    proc = mods["diff"].SO3Diffusion(net, timesteps=100).to(DEV)
    opt = mods["optim"].Adam(net, lr=1e-3)
    x = mods["util"].quat_to_rmat(torch.randn(512, 4, device=DEV))
    g = TrainStepGraph(proc, opt, x.shape, pipeline=True)
    g.step(x)
    opt.param_groups[0]["lr"] = 5e-4
    with pytest.raises(RuntimeError, match="hyper-parameters"):
        g.step(x)
    opt.param_groups[0]["lr"] = 1e-3
    g.step(x)
    g.flush()


def test_given_timesteps_are_clamped_and_never_aliased(mods):
    """(i) p_losses(x, t) with a timestep outside [0, T) must not read outside the tables (the reference raises
    IndexError; the kernels clamp, as so3x_p_mean_t documents) -- the step equals the one with the clamped t, bit for bit;
    (ii) the timesteps train_fwd returns in its carry are a fresh tensor, never the caller's (operator outputs must not alias
    inputs)."""
    B = mods["B"]
    torch.manual_seed(0)
    net = mods["train"].RotPredict(out_type="skewvec", precision="bf16").to(DEV)
    T = 100
    proc = mods["diff"].SO3Diffusion(net, timesteps=T).to(DEV)
    trap_q, _ = proc._tables()
    n = 777
    x0 = B.quat_to_rmat(torch.randn(n, 4, device=DEV))
    t_bad = torch.randint(0, T, (n,), device=DEV)
    t_bad[0], t_bad[5], t_bad[-1] = T, -3, 10 * T
    t_ok = t_bad.clamp(0, T - 1)
    outs = []
    for t in (t_bad, t_ok):
        loss, carry, out = B.train_fwd(net.flat_data(), proc._sched, trap_q, x0, t, seed=1, rng_offset=2, guide_q=proc._guide_q, want_out=True)
        assert carry[1].data_ptr() != t.data_ptr() and torch.equal(carry[1], t_ok)
        outs.append((loss, carry[0], out, B.train_bwd(carry, 17358, T)))
    for a, b in zip(*outs):
        assert torch.isfinite(a).all() and torch.equal(a, b)
    xa, ta, _ = B.q_sample_target(proc._sched, trap_q, x0, t_bad, seed=1, rng_offset=2, guide_q=proc._guide_q)
    xb, tb, _ = B.q_sample_target(proc._sched, trap_q, x0, t_ok, seed=1, rng_offset=2, guide_q=proc._guide_q)
    assert torch.equal(xa, xb) and torch.equal(ta, tb)


def test_sampling_after_graph_training_sees_the_new_weights(mods):
    """a replayed training step updates the parameters in place without touching any Python-side version counter: sampling
    must use the updated weights (the network's flat buffer IS the parameters; nothing cached can go stale)"""
    torch.manual_seed(0)
    base = mods["train"].RotPredict(out_type="skewvec", precision="bf16").to(DEV)
    x = mods["util"].quat_to_rmat(torch.randn(1024, 4, device=DEV))
    net = copy.deepcopy(base)
    proc = mods["diff"].SO3Diffusion(net, timesteps=100).to(DEV)
    from so3x.graphs import TrainStepGraph
    mods["rng"].manual_seed(3)
    before = proc.p_sample_loop((256,))
    opt = mods["optim"].Adam(net, lr=3e-2)
    g = TrainStepGraph(proc, opt, x.shape, warmup=1)
    for _ in range(5):
        g.step(x)
    g.flush()   # the pipelined step is one update behind until flushed
    mods["rng"].manual_seed(3)
    after = proc.p_sample_loop((256,))
    assert float((after - before).abs().max()) > 1e-3
    # and they are the samples of a fresh network that was GIVEN the trained weights
    fresh = mods["train"].RotPredict(out_type="skewvec", precision="bf16").to(DEV)
    fresh.load_state_dict(net.state_dict())
    proc2 = mods["diff"].SO3Diffusion(fresh, timesteps=100).to(DEV)
    mods["rng"].manual_seed(3)
    assert torch.equal(proc2.p_sample_loop((256,)), after)


_DP_GRAPH_WORKER = r'''
import os, sys, json, torch
sys.path.insert(0, sys.argv[1])
mode = sys.argv[2]                           # "eager" | "graph" | "eager-staged" | "graph-pipelined"
kernel = "staged" if mode.endswith(("-staged", "-pipelined")) else "fused"
pipe = True if mode.endswith("-pipelined") else "auto"
mode = mode.split("-")[0]
from so3x import parallel, backend as B, rng, optim
from so3x.so3_train import RotPredict
from so3x.diffusion import SO3Diffusion
from so3x.graphs import TrainStepGraph
ctx = parallel.init()                        # SO3X_DIST_BACKEND=gloo, both ranks on cuda:0 (or world size 1)
torch.manual_seed(100 + ctx.rank)            # different initial weights per rank on purpose: broadcast must fix it
net = RotPredict(out_type="skewvec", precision="bf16").to(ctx.device)
parallel.broadcast_parameters(net, ctx)
start = net.flat_data().clone()
proc = SO3Diffusion(net, timesteps=100).to(ctx.device)   # default quirk_col0=True: global sample 0's row on every shard
proc.rng_counter = torch.zeros(1, dtype=torch.int64, device=ctx.device)
proc.train_step_kernel = kernel
rng.manual_seed(7)
opt = optim.Adam(net, lr=1e-3)
glob = 4096
lo, hi = parallel.shard_range(glob, ctx.rank, ctx.world_size)
proc.index_base = lo
x_all = B.quat_to_rmat(torch.randn(glob, 4, generator=torch.Generator().manual_seed(7)).to(ctx.device))
x = x_all[lo:hi].contiguous()
losses = []
if mode == "graph":
    g = TrainStepGraph(proc, opt, x.shape, warmup=2, ctx=ctx, n_global=glob, pipeline=pipe)
    assert g.fused == (kernel == "fused") and g.pipelined == (pipe is True)
    assert torch.equal(net.flat_data(), start) and int(proc.rng_counter) == 0   # construction leaves no trace
    for _ in range(5):
        losses.append(parallel.mean_scalar(g.step(x).clone(), ctx))
    g.flush()
    gmode = g.mode
else:
    for _ in range(5):
        loss = proc(x)
        opt.zero_grad()
        loss.backward()
        parallel.allreduce_gradients(net, ctx, n_local=hi - lo, n_global=glob, optimizer=opt)
        opt.step()
        losses.append(parallel.mean_scalar(loss.detach(), ctx))
    gmode = "eager"
flat = net.flat_data()
if ctx.world_size > 1:
    both = [torch.zeros_like(flat) for _ in range(ctx.world_size)]
    torch.distributed.all_gather(both, flat)
    assert all(torch.equal(both[0], b) for b in both[1:]), "replicas diverged"
if ctx.rank == 0:
    torch.save({"params": flat.cpu(), "losses": losses, "mode": gmode, "world": ctx.world_size}, sys.argv[3])
parallel.finalize(ctx)
print("OK", ctx.rank, gmode, losses)
'''


def _free_port():
    """a port the kernel just handed out and released (as bench.py does): no fixed numbers that a neighbouring job may hold"""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _launch(tmp_path, mode, world, port, out):
    script = tmp_path / "dp_graph_worker.py"
    script.write_text(_DP_GRAPH_WORKER)
    from conftest import PKG
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world), LOCAL_RANK="0",
               SO3X_DIST_BACKEND="gloo")
    procs = [subprocess.Popen([sys.executable, str(script), PKG, mode, str(out)], env=dict(env, RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = [p.communicate(timeout=900)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0 and "OK" in o, o
    return torch.load(out)


def test_data_parallel_graph_step_two_ranks_on_one_gpu(tmp_path):
    """BASELINE config 4's step, two ranks sharing cuda:0 (collectives over gloo): the graph-replayed data-parallel step
    ([forward + backward] graph, flat-gradient all-reduce, [Adam] graph) keeps the replicas bit-identical and equals the
    eager data-parallel loop bit for bit; and both equal -- to summation-order rounding -- the single-process run on the
    whole batch (noise and timesteps are keyed by the global sample index)."""
    eager2 = _launch(tmp_path, "eager", 2, _free_port(), tmp_path / "e2.pt")
    graph2 = _launch(tmp_path, "graph", 2, _free_port(), tmp_path / "g2.pt")
    assert graph2["mode"] == "split"                                     # gloo cannot be captured: two graphs + eager collective
    assert graph2["losses"] == eager2["losses"] and torch.equal(graph2["params"], eager2["params"])
    # round 3's pipelined stages (on request) against the eager loop on the same three kernels
    eager2s = _launch(tmp_path, "eager-staged", 2, _free_port(), tmp_path / "e2s.pt")
    graph2p = _launch(tmp_path, "graph-pipelined", 2, _free_port(), tmp_path / "g2p.pt")
    assert graph2p["mode"] == "split" and graph2p["losses"] == eager2s["losses"] and torch.equal(graph2p["params"], eager2s["params"])
    graph1 = _launch(tmp_path, "graph", 1, _free_port(), tmp_path / "g1.pt")
    assert graph1["mode"] == "in_graph"
    assert np.allclose(graph1["losses"], graph2["losses"], rtol=1e-4)
    # Adam's first updates are +-lr whatever a gradient's size, so the few gradients that are zero to rounding may move
    # either way; everything else agrees to rounding
    diff = (graph1["params"] - graph2["params"]).abs()
    assert float(diff.median()) < 1e-6 and float((diff > 1e-4).float().mean()) < 0.01


def test_rccl_all_reduce_is_capturable_in_the_training_graph(tmp_path):
    """the in-graph mode needs torch's RCCL process group to accept a collective during stream capture.  One GPU here, so
    world size 1 over the real 'nccl' backend: the capture path (ProcessGroupNCCL under hipGraph capture) is what is
    exercised; the 8-GPU run uses the same code with the communicator spanning the node."""
    code = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=sys.argv[2], WORLD_SIZE='1', RANK='0', LOCAL_RANK='0')
torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1)
from so3x import parallel, backend as B, optim
from so3x.so3_train import RotPredict
from so3x.diffusion import SO3Diffusion
from so3x.graphs import TrainStepGraph
ctx = parallel.Ctx(0, 2, 0, torch.device('cuda:0'))     # pretend world size 2 so that the all-reduce is issued
net = RotPredict(out_type='skewvec', precision='bf16').to('cuda:0')
proc = SO3Diffusion(net, timesteps=100).to('cuda:0')
opt = optim.Adam(net, lr=1e-3)
x = B.quat_to_rmat(torch.randn(1024, 4, device='cuda:0'))
for pipe in ('auto', True):   # the one-kernel step -> reduce -> all-reduce -> Adam, and round 3's pipelined stages
    proc.train_step_kernel = 'fused' if pipe == 'auto' else 'staged'
    g = TrainStepGraph(proc, opt, x.shape, ctx=ctx, allreduce='in_graph', pipeline=pipe)   # ctx says world size 2: the collective is issued
    before = net.flat_data().clone()
    l = [float(g.step(x)) for _ in range(3)]
    g.flush()
    assert g.fused == (pipe == 'auto') and g.pipelined == (pipe is True) and g.mode == 'in_graph'
    assert all(v == v for v in l) and not torch.equal(before, net.flat_data())
dist.destroy_process_group()
print('OK', l)
"""
    from conftest import PKG
    r = subprocess.run([sys.executable, "-c", code, PKG, str(_free_port())], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "OK" in r.stdout, r.stdout + r.stderr


_CAPTURE_FAILURE_WORKER = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
inject = sys.argv[2] == "inject"
from so3x import parallel, backend as B, rng, optim
from so3x.so3_train import RotPredict
from so3x.diffusion import SO3Diffusion
from so3x.graphs import TrainStepGraph
ctx = parallel.init()                        # SO3X_DIST_BACKEND=gloo, both ranks on cuda:0
real = parallel.allreduce_flat
def capture_friendly(flat, ctx_, *a, **k):
    # stands in for a capturable collective (RCCL): inside a stream capture it records a kernel, outside it is the real thing
    if torch.cuda.is_current_stream_capturing():
        return flat.mul_(1.0)
    return real(flat, ctx_, *a, **k)
parallel.allreduce_flat = capture_friendly
torch.manual_seed(5)
net = RotPredict(out_type="skewvec", precision="bf16").to(ctx.device)
parallel.broadcast_parameters(net, ctx)
proc = SO3Diffusion(net, timesteps=100).to(ctx.device)
rng.manual_seed(7)
opt = optim.Adam(net, lr=1e-3)
glob = 2048
lo, hi = parallel.shard_range(glob, ctx.rank, ctx.world_size)
proc.index_base = lo
x = B.quat_to_rmat(torch.randn(glob, 4, generator=torch.Generator().manual_seed(7)).to(ctx.device))[lo:hi].contiguous()
g = TrainStepGraph(proc, opt, x.shape, ctx=ctx, n_global=glob, _assume_capturable=True,
                   _inject_capture_failure=(inject and ctx.rank == 1))
modes = [None, None]
dist.all_gather_object(modes, g.mode)
assert modes[0] == modes[1], modes          # whatever happened on one rank, both replay the same program
if inject:
    assert g.mode == "split", g.mode        # rank 1 failed -> rank 0 (whose capture worked) falls back with it
    for _ in range(4):                      # and the ranks stay in lockstep through real collectives
        g.step(x)
    g.flush()
    flat = net.flat_data()
    both = [torch.zeros_like(flat) for _ in range(2)]
    dist.all_gather(both, flat)
    assert torch.equal(both[0], both[1]) and torch.isfinite(flat).all()
else:
    assert g.mode == "in_graph", g.mode     # control: with no failure anywhere the in-graph form is taken
parallel.finalize(ctx)
print("OK", ctx.rank, g.mode)
'''


@pytest.mark.parametrize("inject", ["inject", "control"])
def test_in_graph_or_split_is_decided_by_all_ranks_together(tmp_path, inject):
    """a per-rank try/except around a capture that contains a collective lets one rank replay
    [.. all-reduce ..] as ONE graph while its peer, whose capture failed, issues the all-reduce eagerly between two graphs -- or
    raises and leaves the peer hanging in the next collective.  TrainStepGraph now takes the decision with one MIN all-reduce of
    'my capture worked'.  Two ranks on this box's one GPU over gloo; the gradient collective is swapped for a stand-in that is
    capturable (records a kernel under capture, is the real all-reduce outside), and rank 1's capture is made to fail."""
    script = tmp_path / "capture_failure_worker.py"
    script.write_text(_CAPTURE_FAILURE_WORKER)
    from conftest import PKG
    port = _free_port()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE="2", LOCAL_RANK="0", SO3X_DIST_BACKEND="gloo")
    procs = [subprocess.Popen([sys.executable, str(script), PKG, inject], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=600)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0 and "OK" in o, o


def test_bench_eight_ranks_dry_run_on_one_gpu():
    """the driver's 8-GPU command has only ever seen world sizes 1 and 2.  `python3 bench.py --gpus 8` as the
    plain command (bench.py starts its eight ranks itself, all on this box's one GPU, collectives over gloo): rendezvous on a
    free port, barriers, max-over-ranks timing, the training leg's one-kernel step with its 8-way gradient all-reduce, ONE
    well-formed line from rank 0, exit code 0 -- inside five minutes.  The throughput is meaningless; the plumbing is the point."""
    import json
    from conftest import ROOT
    env = dict(os.environ, SO3X_DIST_BACKEND="gloo")
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--batch-log2", "15", "--steps", "100", "--warmup", "20",
                        "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 8 and line["ranks_seen"] == 8 and line["backend"] == "gloo" and len(line["devices"]) == 8
    assert line["finite"] and line["value"] > 0 and line["scaling"] == "weak"
    tr = line["train_step"]
    assert "error" not in tr, tr
    assert tr["ranks"] == 8 and tr["global_batch"] == 8 * tr["batch_per_gpu"] and tr["finite"] and tr["one_kernel"]
    assert tr["allreduce_us"] > 0 and tr["mode"].startswith("captured hipGraphs with the eager all-reduce")


# ---------------------------------------------------------------------------------------- prepared-state sampling (round 5)
@pytest.mark.gpu
def test_one_step_per_call_loop_equals_the_chain_and_never_goes_stale(mods):
    """The reference's own way of driving the sampler -- one p_sample call per reverse step with a (1,)-shaped device t
    (so3_test.py:24-31) -- runs from a prepared workspace that is cached across calls (so3x_p_sample_prepare /
    so3x_p_sample_prepared, t read on the device).  It must (a) reproduce the one-launch chain bit for bit, (b) build the
    preparation ONCE for the whole loop, (c) see every kind of parameter update: optimizer steps through the operator (tensor
    version), captured-graph replays (out-of-band epoch), load_state_dict."""
    from so3x import backend as B
    from so3x.graphs import TrainStepGraph
    torch.manual_seed(0)
    net = mods["train"].RotPredict(out_type="skewvec", precision="bf16").to(DEV)
    T = 60
    proc = mods["diff"].SO3Diffusion(net, timesteps=T).to(DEV)
    x0 = mods["util"].quat_to_rmat(torch.randn(2048, 4, device=DEV))

    def loop(x):
        for i in reversed(range(T)):
            x = proc.p_sample(x, torch.full((1,), i, device=DEV, dtype=torch.long))
        return x

    def chain(x):
        _, trap_p = proc._tables()
        return B.p_sample_chain(net.flat_params_nograd(), proc._sched, trap_p, x, T - 1, T, seed=mods["rng"].seed(), rng_offset=0,
                                precision=net.precision_code, guide_p=proc._guide_p)

    calls = {"n": 0}
    real = B.p_sample_prepare

    def counting(*a, **k):
        calls["n"] += 1
        return real(*a, **k)
    B.p_sample_prepare = counting
    try:
        mods["rng"].manual_seed(9)
        a = loop(x0)                      # every call draws offset o_k = k * T and the kernel adds t: distinct streams per step
        assert calls["n"] == 1            # (b)
        # (a): the same steps as separate one-step launches of the unprepared entry point, same offsets
        mods["rng"].manual_seed(9)
        x = x0
        _, trap_p = proc._tables()
        for k, i in enumerate(reversed(range(T))):
            x = B.p_sample_chain(net.flat_params_nograd(), proc._sched, trap_p, x, i, 1, seed=mods["rng"].seed(), rng_offset=k * T,
                                 precision=net.precision_code, guide_p=proc._guide_p)
        assert torch.equal(a, x)
        # int t and [B]-shaped t take the prepared path too and agree
        mods["rng"].manual_seed(9)
        b1 = proc.p_sample(x0, 17)
        mods["rng"].manual_seed(9)
        b2 = proc.p_sample(x0, torch.full((2048,), 17, device=DEV, dtype=torch.long))
        mods["rng"].manual_seed(9)
        b3 = proc.p_sample(x0, torch.full((1,), 17, device=DEV, dtype=torch.long))
        assert torch.equal(b1, b2) and torch.equal(b1, b3) and calls["n"] == 1
        # (c) optimizer step through the operator
        opt = mods["optim"].Adam(net, lr=3e-2)
        loss = proc(x0)
        loss.backward()
        opt.step()
        mods["rng"].manual_seed(9)
        c = proc.p_sample(x0, 17)
        assert calls["n"] == 2 and float((c - b1).abs().max()) > 1e-4
        mods["rng"].manual_seed(9)
        _, trap_p = proc._tables()
        want = B.p_sample_chain(net.flat_params_nograd(), proc._sched, trap_p, x0, 17, 1, seed=mods["rng"].seed(), rng_offset=0,
                                precision=net.precision_code, guide_p=proc._guide_p)
        assert torch.equal(c, want)
        # (c) graph replays
        del loss
        opt.zero_grad(set_to_none=True)
        g = TrainStepGraph(proc, opt, x0.shape, warmup=1)
        for _ in range(3):
            g.step(x0)
        g.flush()
        mods["rng"].manual_seed(9)
        d = proc.p_sample(x0, 17)
        mods["rng"].manual_seed(9)
        want = B.p_sample_chain(net.flat_params_nograd(), proc._sched, trap_p, x0, 17, 1, seed=mods["rng"].seed(), rng_offset=0,
                                precision=net.precision_code, guide_p=proc._guide_p)
        assert torch.equal(d, want) and float((d - c).abs().max()) > 1e-4
        # (c) load_state_dict
        sd = {k: v * 0.5 for k, v in net.state_dict().items()}
        net.load_state_dict(sd)
        mods["rng"].manual_seed(9)
        e = proc.p_sample(x0, 17)
        mods["rng"].manual_seed(9)
        want = B.p_sample_chain(net.flat_params_nograd(), proc._sched, trap_p, x0, 17, 1, seed=mods["rng"].seed(), rng_offset=0,
                                precision=net.precision_code, guide_p=proc._guide_p)
        assert torch.equal(e, want)
        # (d) a write torch's version counters do NOT see (EMA / weight averaging through `p.data`): found by the buffer's
        # fingerprint at the next chain start, after train() / eval(), after an idle gap, or on request
        built = calls["n"]
        for p_ in net.parameters():
            p_.data.mul_(0.9)
        mods["rng"].manual_seed(9)
        f = proc.p_sample(x0, torch.full((1,), T - 1, device=DEV, dtype=torch.long).item())   # host-known t = T - 1: a chain start
        mods["rng"].manual_seed(9)
        want = B.p_sample_chain(net.flat_params_nograd(), proc._sched, trap_p, x0, T - 1, 1, seed=mods["rng"].seed(), rng_offset=0,
                                precision=net.precision_code, guide_p=proc._guide_p)
        assert calls["n"] == built + 1 and torch.equal(f, want)
        for p_ in net.parameters():
            p_.data.mul_(0.9)
        proc.eval()                                                      # the mode switch arms the comparison for the next call
        mods["rng"].manual_seed(9)
        g_ = proc.p_sample(x0, torch.full((1,), 17, device=DEV, dtype=torch.long))
        mods["rng"].manual_seed(9)
        want = B.p_sample_chain(net.flat_params_nograd(), proc._sched, trap_p, x0, 17, 1, seed=mods["rng"].seed(), rng_offset=0,
                                precision=net.precision_code, guide_p=proc._guide_p)
        assert calls["n"] == built + 2 and torch.equal(g_, want)
        for p_ in net.parameters():
            p_.data.mul_(0.9)
        proc.invalidate_sampling_cache()
        proc.p_sample(x0, 17)
        assert calls["n"] == built + 3
        proc.p_sample(x0, 16)                                            # unchanged parameters: no rebuild, whatever was compared
        assert calls["n"] == built + 3
    finally:
        B.p_sample_prepare = real
    # the device-side timestep is clamped into the tables, never read outside them
    ws = B.p_sample_prepare(net.flat_params_nograd(), proc._sched, trap_p, net.precision_code, guide_p=proc._guide_p)
    hi = B.p_sample_prepared(ws, proc._sched, trap_p, x0, 0, 1, t_dev=torch.full((1,), 10 ** 6, device=DEV, dtype=torch.long), seed=1,
                             precision=net.precision_code, guide_p=proc._guide_p)
    top = B.p_sample_prepared(ws, proc._sched, trap_p, x0, 0, 1, t_dev=torch.full((1,), T - 1, device=DEV, dtype=torch.long), seed=1,
                              precision=net.precision_code, guide_p=proc._guide_p)
    assert torch.equal(hi, top)


def test_direct_backward_is_the_engines_backward(mods):
    """The reference loop's `loss.backward()` (so3_train.py:71) skips the autograd engine on the fast path; everything the engine
    would have done still holds: the same gradient bit for bit, a second backward refused, derived losses and accumulation
    through the engine, tensor hooks and optimizer step hooks honoured."""
    B, rng = mods["B"], mods["rng"]
    torch.manual_seed(3)
    net = mods["train"].RotPredict(out_type="skewvec", precision="bf16").to(DEV)
    proc = mods["diff"].SO3Diffusion(net, timesteps=1000).to(DEV)
    x0 = B.quat_to_rmat(torch.randn(4096, 4, device=DEV))

    def grads(how):
        rng.manual_seed(11)
        net.zero_grad(set_to_none=True)
        loss = proc(x0)
        assert "backward" in loss.__dict__ and "FusedSkewvecLoss" in type(loss.grad_fn).__name__
        how(loss)
        return float(loss), net.gather_flat_grad().clone()

    l_direct, g_direct = grads(lambda l: l.backward())
    l_engine, g_engine = grads(lambda l: l.backward(torch.ones((), device=DEV)))
    assert l_direct == l_engine and torch.equal(g_direct, g_engine)
    assert all(p.grad is not None and p.grad.shape == p.shape for p in net.parameters())
    _, g_twice = grads(lambda l: (l * 2).backward())                 # a derived loss is a plain tensor: the engine's path
    assert torch.equal(g_twice, 2 * g_direct)

    rng.manual_seed(11)
    net.zero_grad(set_to_none=True)
    loss = proc(x0)
    loss.backward()
    with pytest.raises(RuntimeError, match="already consumed"):
        loss.backward()
    rng.manual_seed(11)
    proc(x0).backward()                                               # no zero_grad: accumulated (through the engine)
    assert torch.allclose(net.gather_flat_grad(), 2 * g_direct, rtol=1e-6, atol=0)

    # torch.autograd.grad and backward(inputs=...) get per-parameter gradients from the engine and touch no other .grad
    rng.manual_seed(11)
    net.zero_grad(set_to_none=True)
    ps = list(net.parameters())
    gs = torch.autograd.grad(proc(x0), ps)
    assert all(p.grad is None for p in ps) and torch.equal(torch.cat([g_.reshape(-1) for g_ in gs]), g_direct)
    rng.manual_seed(11)
    loss = proc(x0)
    loss.backward(inputs=[ps[2]])
    assert ps[2].grad is not None and all(p.grad is None for i, p in enumerate(ps) if i != 2)
    n0 = ps[0].numel() + ps[1].numel()
    assert torch.equal(ps[2].grad.reshape(-1), g_direct[n0:n0 + ps[2].numel()])
    net.zero_grad(set_to_none=True)
    with pytest.raises(RuntimeError, match="bind it to a name"):
        proc(x0).backward(inputs=[ps[2]])
    # a loss nobody differentiates goes away with its last reference (no cycle through its `backward` attribute)
    import gc
    import weakref
    gc.disable()
    try:
        loss = proc(x0)
        w_ = weakref.ref(loss)
        del loss
        assert w_() is None
    finally:
        gc.enable()

    seen = []
    h = net.net[0].weight.register_hook(lambda g: seen.append(g.shape))
    _, g_hooked = grads(lambda l: l.backward())
    h.remove()
    assert seen == [net.net[0].weight.shape] and torch.equal(g_hooked, g_direct)

    opt = mods["optim"].Adam(net, lr=1e-3)
    calls = []
    h = opt.register_step_post_hook(lambda o, a, k: calls.append("post"))
    before = net.flat_data().clone()
    opt.step()
    h.remove()
    assert calls == ["post"] and not torch.equal(before, net.flat_data())
    opt.zero_grad()
    assert all(p.grad is None for p in net.parameters()) and net.flat_grad() is None
