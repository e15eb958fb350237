"""PlaneNet on the hand-written kernels (reference models.py:185-210, aircraft_rotate.py:64-117):
so3x_planenet_fwd / so3x_planenet_bwd against values the reference's own modules produced (tools/make_golden.py planenet).

Two fixtures: `planenet.npz` -- a small network (dim 32, 4 heads, 2 layers) with its full state_dict, every intermediate and every
parameter gradient; `planenet_full.npz` -- the aircraft task's own width (dim 512, 4 heads, 4 layers, 12.9 M parameters) at 24, 256
and 2048 points, whose weights are rebuilt here from the same seeds (checksums in the fixture pin them)."""
import numpy as np
import pytest
import torch

DEV = "cuda:0"


def dev(a, dtype=torch.float32):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV).to(dtype)


def planenet_perturb(net, seed):
    """tools/make_golden.py:planenet_perturb, verbatim: a seeded nudge of every parameter (a fresh nn.TransformerEncoder's layers
    are deep copies of one layer; the nudge makes them differ)"""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for _, p in sorted(net.named_parameters()):
            p.add_(torch.randn(p.shape, generator=g) * (0.05 * float(p.abs().mean()) + 1e-3))


def small_net(golden, precision="fp32"):
    from so3x.models import PlaneNet
    g = golden["planenet"]
    net = PlaneNet(dim=int(g["dim"]), heads=int(g["heads"]), layers=int(g["layers"]), precision=precision, dropout=0.0)
    sd = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd_")}
    assert list(net.state_dict().keys()) == list(sd.keys())                      # the reference's checkpoint keys, in its order
    net.load_state_dict(sd)
    return net, g


def full_net(golden, precision):
    from so3x.models import PlaneNet
    g = golden["planenet_full"]
    torch.manual_seed(21)
    net = PlaneNet(dim=int(g["dim"]), heads=int(g["heads"]), layers=int(g["layers"]), precision=precision, dropout=0.0)
    planenet_perturb(net, 5)
    for k, v in net.state_dict().items():                                       # the weights ARE the ones the fixture was made with
        chk = g["chk_" + k]
        v64 = v.double()
        assert abs(float(v64.sum()) - chk[0]) <= 1e-9 * max(1.0, abs(chk[0])) and abs(float(v64.norm()) - chk[1]) <= 1e-9 * chk[1], k
    return net, g


# ------------------------------------------------------------------------------------------------ CPU: layout and host logic
def test_flat_parameter_layout_is_the_state_dict_order(golden):
    """the kernels address ONE flat buffer by offsets computed from (dim, heads, layers, ffn): the module's parameters must sit in
    it in state_dict order, and the C library must agree about the count"""
    from so3x import backend as B
    net, g = small_net(golden)
    flat = net.flat_data()
    assert flat.numel() == sum(p.numel() for p in net.parameters()) == B.planenet_param_count(32, 4, 2, 2048)
    off = 0
    for k, v in net.state_dict().items():
        assert torch.equal(flat[off:off + v.numel()].view(v.shape), v), k
        assert v.data_ptr() == flat.data_ptr() + 4 * off, k
        off += v.numel()
    assert B.planenet_param_count(512, 4, 4) == 12_941_060
    with pytest.raises(ValueError):
        B.planenet_param_count(30, 4, 2)       # dim % heads != 0


def test_torch_module_composition_is_the_reference(golden):
    """forward_torch (the torch modules the state_dict keys come from) reproduces the reference's blocks: the fixture and the
    module tree agree, so a kernel that matches the fixture matches the reference"""
    net, g = small_net(golden)
    net.eval()
    with torch.no_grad():
        out = net.forward_torch(torch.from_numpy(g["x"]), torch.from_numpy(g["t"]))
    assert np.abs(out.numpy() - g["out"]).max() < 1e-5


def test_cpu_tensors_are_refused(golden):
    from so3x.backend import So3xError
    net, g = small_net(golden)
    net.eval()
    with pytest.raises(So3xError):
        net(torch.from_numpy(g["x"]), torch.from_numpy(g["t"]))


def philox4x32_10(seed, ctr_lo, ctr_hi):
    """Philox4x32-10 as csrc/so3x_math.hpp runs it (key = seed, counter = (ctr_lo, ctr_hi)), vectorised over ctr_lo: four uint32 words"""
    M = np.uint64(0xFFFFFFFF)
    ctr_lo = np.asarray(ctr_lo, dtype=np.uint64)
    k0, k1 = np.uint64(seed) & M, np.uint64(seed) >> np.uint64(32)
    c0, c1 = ctr_lo & M, ctr_lo >> np.uint64(32)
    c2 = np.full_like(c0, np.uint64(ctr_hi) & M)
    c3 = np.full_like(c0, np.uint64(ctr_hi) >> np.uint64(32))
    for _ in range(10):
        p0, p1 = np.uint64(0xD2511F53) * c0, np.uint64(0xCD9E8D57) * c2
        c0, c1, c2, c3 = (p1 >> np.uint64(32)) ^ c1 ^ k0, p1 & M, (p0 >> np.uint64(32)) ^ c3 ^ k1, p0 & M
        k0, k1 = (k0 + np.uint64(0x9E3779B9)) & M, (k1 + np.uint64(0xBB67AE85)) & M
    return np.stack([c0, c1, c2, c3], axis=-1)


def dropout_mask(n, p, seed, offset, layer, site):
    """include/so3x.h: element e keeps its value iff 16-bit piece (e & 7) of Philox(seed; (e >> 3, offset << 8 | 4 layer + site)),
    words in order, low half first, is >= floor(p 2^16)"""
    thr = min(int(float(np.float32(p)) * 65536.0), 65535)
    w = philox4x32_10(seed, np.arange((n + 7) // 8, dtype=np.uint64), (offset << 8) | (4 * layer + site))       # [calls, 4]
    halves = np.stack([w & np.uint64(0xFFFF), w >> np.uint64(16)], axis=-1).reshape(-1)[:n]                        # x.lo, x.hi, y.lo, ...
    return halves >= np.uint64(thr)


def test_philox_known_answer():
    """Random123's known-answer vectors for philox4x32-10: the emulation the dropout parity test rests on is the generator it says"""
    out = philox4x32_10(0, [0], 0)[0]
    assert [int(v) for v in out] == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    full = (1 << 64) - 1
    out = philox4x32_10(full, [full], full)[0]
    assert [int(v) for v in out] == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    m = dropout_mask(1 << 20, 0.1, 3, 7, 2, 1)
    assert abs(float(m.mean()) - 0.9) < 2e-3


def planenet_with_masks(net, x, t, masks, keep):
    """PlaneNet's forward (reference models.py:198-210 over nn.TransformerEncoderLayer's post-norm arithmetic,
    torch/nn/modules/transformer.py) in float64 on the CPU with GIVEN dropout masks: masks[l] = (attention [B, H, P, P],
    block 1 [B, P, d], feed-forward [B, P, ffn], block 2 [B, P, d]) or None.  Test infrastructure."""
    import torch.nn.functional as Fn
    x_emb = net.position_siren(x)
    t_emb = net.time_embedding(t)
    h = torch.cat((x_emb, t_emb[:, None, :].expand(x_emb.shape)), dim=2)
    B, P, d = h.shape
    H = net.heads
    for l, layer in enumerate(net.encoder.layers):
        ma, m1, mf, m2 = masks[l] if masks is not None else (None,) * 4
        qkv = Fn.linear(h, layer.self_attn.in_proj_weight, layer.self_attn.in_proj_bias)
        q, k, v = (z.reshape(B, P, H, d // H).transpose(1, 2) for z in qkv.split(d, dim=2))
        pr = torch.softmax(q @ k.transpose(2, 3) / (d // H) ** 0.5, dim=-1)
        if ma is not None:
            pr = pr * ma / keep
        o = (pr @ v).transpose(1, 2).reshape(B, P, d)
        y = layer.self_attn.out_proj(o)
        if m1 is not None:
            y = y * m1 / keep
        x1 = layer.norm1(h + y)
        f = torch.relu(layer.linear1(x1))
        if mf is not None:
            f = f * mf / keep
        y2 = layer.linear2(f)
        if m2 is not None:
            y2 = y2 * m2 / keep
        h = layer.norm2(x1 + y2)
    return net.out_net(h)


# ------------------------------------------------------------------------------------------------ GPU: fp32 form vs the reference
def rel(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


@pytest.mark.gpu
def test_small_forward_vs_reference(golden):
    net, g = small_net(golden)
    net = net.to(DEV).eval()
    with torch.no_grad():
        out, enc = net(dev(g["x"]), dev(g["t"], torch.int64), want_encoding=True)
    # fp32 through a SIREN with |pre-activation| up to ~100 (ulp 7.6e-6) and two post-norm layers: the reference's own fp32 run
    # sits this far from its float64 run (fixture: out64 / encoding64)
    e_ref = max(np.abs(g["encoding"] - g["encoding64"]).max(), 1e-6)
    assert np.abs(enc.cpu().numpy() - g["encoding64"]).max() < max(2e-5, 3 * e_ref)
    assert np.abs(out.cpu().numpy() - g["out64"]).max() < max(1e-5, 3 * np.abs(g["out"] - g["out64"]).max())
    assert np.abs(out.cpu().numpy() - g["out"]).max() < 2e-5
    # the plain call returns the same numbers
    with torch.no_grad():
        assert torch.equal(net(dev(g["x"]), dev(g["t"], torch.int64)), out)


@pytest.mark.gpu
def test_small_backward_vs_reference_autograd(golden):
    net, g = small_net(golden)
    net = net.to(DEV).train()      # dropout 0: training mode is accepted
    out = net(dev(g["x"]), dev(g["t"], torch.int64))
    (out * dev(g["dout"])).sum().backward()
    worst = {}
    for k, p in net.named_parameters():
        want = g["grad_" + k]
        assert p.grad is not None and p.grad.shape == want.shape, k
        worst[k] = rel(p.grad.cpu().numpy(), want)
    bad = {k: v for k, v in worst.items() if v > 2e-4}
    assert not bad, bad
    # deterministic: a second backward gives the same bits
    flat1 = net.flat_grad().clone()
    net.zero_grad(set_to_none=True)
    (net(dev(g["x"]), dev(g["t"], torch.int64)) * dev(g["dout"])).sum().backward()
    assert torch.equal(net.flat_grad(), flat1)


@pytest.mark.gpu
@pytest.mark.parametrize("P", [24, 256, 2048])
def test_full_width_fp32_vs_reference(golden, P):
    net, g = full_net(golden, "fp32")
    net = net.to(DEV).train()
    tag = f"P{P}_"
    x, t, dout = dev(g[tag + "x"]), dev(g[tag + "t"], torch.int64), dev(g[tag + "dout"])
    with torch.no_grad():
        out_e, enc = net(x, t, want_encoding=True)
    ends = torch.cat((enc[:, :4], enc[:, -4:]), 1).cpu().numpy()
    assert np.abs(ends - g[tag + "encoding_ends"]).max() < 1e-4          # LayerNorm outputs, O(1) entries
    assert np.abs(out_e.cpu().numpy() - g[tag + "out"]).max() < 2e-5
    out = net(x, t)
    assert torch.equal(out.detach(), out_e)
    (out * dout).sum().backward()
    for k, p in net.named_parameters():
        gs, pick = g[tag + "gsum_" + k], g[tag + "gpick_" + k]
        flat = p.grad.reshape(-1)
        stride = max(1, flat.numel() // 64)
        got = flat[::stride][:64].cpu().numpy()
        scale = gs[1] / np.sqrt(flat.numel())                              # rms entry of this gradient
        assert np.abs(got - pick).max() < 2e-3 * max(scale, np.abs(pick).max()), k
        assert abs(float(flat.double().norm()) - gs[1]) < 1e-3 * gs[1], k


@pytest.mark.gpu
def test_ragged_and_empty_shapes(golden):
    """points not a multiple of any tile, one cloud, zero clouds"""
    from so3x import backend as B
    net, g = small_net(golden)
    net = net.to(DEV).eval()
    cpu = small_net(golden)[0].eval()
    gen = torch.Generator().manual_seed(3)
    for Bn, P in ((1, 1), (3, 67), (2, 130)):
        x = torch.randn(Bn, P, 3, generator=gen) * 0.5
        t = torch.randint(0, 1000, (Bn,), generator=gen)
        with torch.no_grad():
            want = cpu.forward_torch(x, t)
            got = net(x.to(DEV), t.to(DEV))
        assert float((got.cpu() - want).abs().max()) < 3e-5, (Bn, P)
    with torch.no_grad():
        assert net(torch.zeros(0, 5, 3, device=DEV), torch.zeros(0, dtype=torch.long, device=DEV)).shape == (0, 3)


# ------------------------------------------------------------------------------------------------ GPU: bf16 form
@pytest.mark.gpu
@pytest.mark.parametrize("P", [256, 2048])
def test_full_width_bf16_forward_vs_reference(golden, P):
    """bf16 operands AND bf16 activations through four post-norm layers: 3e-2 of the output scale, and the
    encoder output (LayerNorm rows, entries O(1)) within 6e-2 absolute with a median error far below"""
    net, g = full_net(golden, "bf16")
    net = net.to(DEV).eval()
    tag = f"P{P}_"
    x, t = dev(g[tag + "x"]), dev(g[tag + "t"], torch.int64)
    with torch.no_grad():
        out, enc = net(x, t, want_encoding=True)
    want = g[tag + "out"]
    assert np.abs(out.cpu().numpy() - want).max() < 3e-2 * max(1.0, np.abs(want).max())
    ends = torch.cat((enc[:, :4], enc[:, -4:]), 1).cpu().numpy()
    err = np.abs(ends - g[tag + "encoding_ends"])
    assert err.max() < 1e-1 and np.median(err) < 1e-2, (err.max(), np.median(err))


@pytest.mark.gpu
def test_bf16_forward_matches_the_fp32_form_on_odd_shapes(golden):
    """points = 64 and 192 (the last attention block half empty), token counts that are not a multiple of the GEMM tile (pad rows)"""
    net32, _ = full_net(golden, "fp32")
    net16, _ = full_net(golden, "bf16")
    net32, net16 = net32.to(DEV).eval(), net16.to(DEV).eval()
    gen = torch.Generator().manual_seed(11)
    for Bn, P in ((1, 64), (3, 192), (5, 320)):
        x = (torch.randn(Bn, P, 3, generator=gen) * 0.5).to(DEV)
        t = torch.randint(0, 1000, (Bn,), generator=gen).to(DEV)
        with torch.no_grad():
            a, ea = net32(x, t, want_encoding=True)
            b, eb = net16(x, t, want_encoding=True)
        assert float((a - b).abs().max()) < 3e-2 * max(1.0, float(a.abs().max())), (Bn, P)
        assert float((ea - eb).abs().max()) < 1e-1 and float((ea - eb).abs().median()) < 1e-2, (Bn, P)
    with pytest.raises(Exception):
        net16(torch.zeros(1, 24, 3, device=DEV), torch.zeros(1, dtype=torch.long, device=DEV))   # 24 points: not this form's shape


@pytest.mark.gpu
def test_bf16_large_batch_kernels_equal_the_small_batch_kernels(golden):
    """65,536 tokens take the persistent 256 x 256-tile GEMM; the same clouds four at a time take the 128 x 128-tile one (the
    form the fixtures pin).  Both sum every output element over k in the same order, so the encoder outputs must agree to the last
    bf16 digit on almost every entry and the predictions to fp32 rounding."""
    net, _ = full_net(golden, "bf16")
    net = net.to(DEV).eval()
    gen = torch.Generator(device=DEV).manual_seed(5)
    x = torch.randn(32, 2048, 3, device=DEV, generator=gen) * 0.5
    t = torch.randint(0, 1000, (32,), device=DEV, generator=gen)
    from so3x import backend as B
    with torch.no_grad():
        # (with a stash: the plain kernel sequence -- without one, inference at this size folds its LayerNorms, which is a different
        #  rounding and has its own test below)
        big, _, ebig = B.planenet_fwd(net.flat_data(), x, t, *net.cfg, want_stash=True, want_encoding=True)
        for i in range(0, 32, 8):
            small, esmall = net(x[i:i + 4], t[i:i + 4], want_encoding=True)
            assert float((big[i:i + 4] - small).abs().max()) < 1e-5, i
            d = (ebig[i:i + 4] - esmall).abs()
            assert float(d.max()) < 4e-2 and float((d > 0).float().mean()) < 1e-3, (i, float(d.max()), float((d > 0).float().mean()))
    assert torch.isfinite(big).all()


@pytest.mark.gpu
@pytest.mark.parametrize("P", [256, 2048])
def test_full_width_bf16_backward_vs_reference_autograd(golden, P):
    """bf16 operands, activations AND activation gradients; parameter gradients accumulated in fp32.  Against the reference's
    fp32 autograd: every parameter's gradient within 2.5 % in norm and its 64 sampled entries within 2.5 % of the larger of the
    gradient's rms entry and the largest sampled entry (measured: 0.2-1.6 %) -- except the SIREN's two matrices, the deepest tensors
    of the backward (the activation gradient has been through all four layers in bf16 and the product sums 65,536 tokens of bf16
    operands): one of their 64 sampled entries reaches 2.9 % at 2048 points (norms: 0.02 %); their entry gate is 3.5 %."""
    net, g = full_net(golden, "bf16")
    net = net.to(DEV).train()
    tag = f"P{P}_"
    x, t, dout = dev(g[tag + "x"]), dev(g[tag + "t"], torch.int64), dev(g[tag + "dout"])
    out = net(x, t)
    (out * dout).sum().backward()
    bad = {}
    for k, p in net.named_parameters():
        gs, pick = g[tag + "gsum_" + k], g[tag + "gpick_" + k]
        flat = p.grad.reshape(-1)
        assert torch.isfinite(flat).all(), k
        stride = max(1, flat.numel() // 64)
        got = flat[::stride][:64].cpu().numpy()
        scale = max(gs[1] / np.sqrt(flat.numel()), np.abs(pick).max())
        norm = gs[1]
        if k == "out_net.0.pool.0.bias":
            # d bpool = sum_p e_p w_p (1 - w_p) with sum_p e_p w_p = 0 identically: a cancelling sum, whose bf16 noise does not
            # cancel.  Its natural scale is the gradient of the pooling weight it sits beside (same sum, weighted by x_p = O(1)).
            scale = norm = g[tag + "gsum_out_net.0.pool.0.weight"][1]
        e_pick = float(np.abs(got - pick).max() / scale)
        e_norm = abs(float(flat.double().norm()) - gs[1]) / norm
        if e_pick > (3.5e-2 if k.startswith("position_siren.") else 2.5e-2) or e_norm > 2.5e-2:
            bad[k] = (e_pick, e_norm)
    assert not bad, bad
    # deterministic: fixed-order reductions, no atomics
    g1 = net.flat_grad().clone()
    net.zero_grad(set_to_none=True)
    (net(x, t) * dout).sum().backward()
    assert torch.equal(net.flat_grad(), g1)


@pytest.mark.gpu
def test_bf16_large_batch_backward_equals_the_small_batch_backward(golden):
    """From 32,768 tokens on the weight gradients take the 256 x 256-tile token-major product (k_gemm_tn256: other tiles, other
    cuts of the token axis, its own bias column sums); the same clouds four at a time take the 128 x 128-tile one, which the
    fixtures pin.  The gradient of a sum over clouds is the sum of the groups' gradients: the two must agree to the rounding of the
    activations' bf16 forms and of fp32 sums taken in another order."""
    net, _ = full_net(golden, "bf16")
    net = net.to(DEV).train()
    gen = torch.Generator(device=DEV).manual_seed(9)
    x = torch.randn(16, 2048, 3, device=DEV, generator=gen) * 0.5
    t = torch.randint(0, 1000, (16,), device=DEV, generator=gen)
    dout = torch.randn(16, 3, device=DEV, generator=gen)
    net.zero_grad(set_to_none=True)
    (net(x, t) * dout).sum().backward()
    big = {k: p.grad.clone() for k, p in net.named_parameters()}
    small = {k: torch.zeros_like(v) for k, v in big.items()}
    for i in range(0, 16, 4):
        net.zero_grad(set_to_none=True)
        (net(x[i:i + 4], t[i:i + 4]) * dout[i:i + 4]).sum().backward()
        for k, p in net.named_parameters():
            small[k] += p.grad
    for k in big:
        assert torch.isfinite(big[k]).all(), k
        rel = float((big[k] - small[k]).norm() / (small[k].norm() + 1e-30))
        assert rel < 2e-3, (k, rel)
    # deterministic
    g1 = net.flat_grad().clone()
    net.zero_grad(set_to_none=True)
    (net(x[12:16], t[12:16]) * dout[12:16]).sum().backward()
    assert torch.equal(net.flat_grad(), g1)


@pytest.mark.gpu
def test_bf16_backward_matches_the_fp32_form_on_odd_shapes(golden):
    """pad rows (tokens not a multiple of 128) and a half-empty last attention block must not leak into any gradient"""
    net32, _ = full_net(golden, "fp32")
    net16, _ = full_net(golden, "bf16")
    net32, net16 = net32.to(DEV).train(), net16.to(DEV).train()
    gen = torch.Generator().manual_seed(12)
    for Bn, P in ((1, 64), (3, 192)):
        x = (torch.randn(Bn, P, 3, generator=gen) * 0.5).to(DEV)
        t = torch.randint(0, 1000, (Bn,), generator=gen).to(DEV)
        dout = torch.randn(Bn, 3, generator=gen).to(DEV)
        for net in (net32, net16):
            net.zero_grad(set_to_none=True)
            (net(x, t) * dout).sum().backward()
        ref = dict(net32.named_parameters())
        for (k, a), (_, b) in zip(net32.named_parameters(), net16.named_parameters()):
            # (the pooling bias' gradient is a cancelling sum: measured against the pooling weight's gradient, see above)
            scale = ref["out_net.0.pool.0.weight"].grad.norm() if k == "out_net.0.pool.0.bias" else a.grad.norm()
            rel = float((a.grad - b.grad).norm() / (scale + 1e-30))
            assert rel < 5e-2 and torch.isfinite(b.grad).all(), (Bn, P, k, rel)


# ------------------------------------------------------------------------------------------------ GPU: training-mode dropout
@pytest.mark.gpu
def test_training_mode_dropout_vs_the_reference_arithmetic_with_the_same_masks(golden):
    """The reference trains PlaneNet in training mode (aircraft_rotate.py:66) on nn.TransformerEncoderLayer's default dropout 0.1.
    The kernels' masks are counter-based (include/so3x.h); torch's own stream cannot be matched, so parity is against the
    reference's arithmetic in float64 WITH THESE MASKS: output and every parameter gradient.  Also: a manual_seed reproduces the
    masks, consecutive forwards draw fresh ones, eval mode has none."""
    import copy
    from so3x import rng
    from so3x.models import PlaneNet
    g = golden["planenet"]
    dim, heads, layers = int(g["dim"]), int(g["heads"]), int(g["layers"])
    net = PlaneNet(dim=dim, heads=heads, layers=layers, precision="fp32", dropout=0.1)
    net.load_state_dict({k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd_")})
    ref = copy.deepcopy(net).double()
    x64, t64 = torch.from_numpy(g["x"]).double(), torch.from_numpy(g["t"]).long()
    Bn, P = x64.shape[0], x64.shape[1]
    ref.eval()
    with torch.no_grad():   # the emulation without masks IS the torch modules' eval forward
        assert float((planenet_with_masks(ref, x64, t64, None, 1.0) - ref.forward_torch(x64, t64)).abs().max()) < 1e-12
    net = net.to(DEV).train()
    x, t = dev(g["x"]), dev(g["t"], torch.int64)
    seed, p32 = 1234, float(np.float32(0.1))
    keep = 1.0 - p32
    rng.manual_seed(seed)                      # offset 0 for the next forward
    out = net(x, t)
    dout = torch.from_numpy(np.random.default_rng(0).standard_normal((Bn, 3))).to(DEV).float()
    net.zero_grad(set_to_none=True)
    (out * dout).sum().backward()
    ffn = net.ffn
    masks = []
    for l in range(layers):
        masks.append(tuple(torch.from_numpy(dropout_mask(n, 0.1, seed, 0, l, site).reshape(shape)).double()
                           for site, (n, shape) in enumerate(((Bn * heads * P * P, (Bn, heads, P, P)), (Bn * P * dim, (Bn, P, dim)),
                                                              (Bn * P * ffn, (Bn, P, ffn)), (Bn * P * dim, (Bn, P, dim))))))
    ref.zero_grad(set_to_none=True)
    out_ref = planenet_with_masks(ref, x64, t64, masks, keep)
    (out_ref * dout.double().cpu()).sum().backward()
    assert rel(out.detach().cpu().numpy(), out_ref.detach().numpy()) < 2e-5
    no_mask = planenet_with_masks(ref, x64, t64, None, 1.0).detach().numpy()
    assert rel(out.detach().cpu().numpy(), no_mask) > 1e-2           # (the masks do something: the test has power)
    for (k, pk), (_, pr) in zip(net.named_parameters(), ref.named_parameters()):
        a, b = pk.grad.cpu().numpy(), pr.grad.numpy()
        assert np.abs(a - b).max() <= 5e-4 * max(np.abs(b).max(), 1e-6) + 1e-7, k
    # the same seed and offset: the same masks, bit for bit; the next forward: fresh ones; eval: none
    rng.manual_seed(seed)
    with torch.no_grad():
        again = net(x, t)
        other = net(x, t)
    assert torch.equal(again, out.detach()) and not torch.equal(other, again)
    net.eval()
    with torch.no_grad():
        assert rel(net(x, t).cpu().numpy(), no_mask) < 2e-5


@pytest.mark.gpu
def test_bf16_dropout_draws_the_masks_of_the_fp32_form(golden):
    """Training mode at the aircraft task's width: the bf16 kernels (dropout in the GEMM epilogues, keep bits inside the attention
    kernels) against the exact-fp32 kernels -- pinned above on the reference's arithmetic -- on the SAME (seed, offset): output and
    every parameter gradient to bf16 accuracy, on a small-batch shape (128-wide tiles), on one with pad rows, and at 24 x 2048
    points, where every product runs on the persistent 256-wide kernel (its own epilogue code)."""
    from so3x import rng
    net32, _ = full_net(golden, "fp32")
    net16, _ = full_net(golden, "bf16")
    for net in (net32, net16):
        net.dropout = 0.1
    net32, net16 = net32.to(DEV).train(), net16.to(DEV).train()
    gen = torch.Generator().manual_seed(13)
    for Bn, P in ((2, 128), (3, 192), (24, 2048)):
        x = (torch.randn(Bn, P, 3, generator=gen) * 0.5).to(DEV)
        t = torch.randint(0, 1000, (Bn,), generator=gen).to(DEV)
        dout = torch.randn(Bn, 3, generator=gen).to(DEV)
        outs = []
        for net in (net32, net16):
            rng.manual_seed(99)
            net.zero_grad(set_to_none=True)
            out = net(x, t)
            (out * dout).sum().backward()
            outs.append(out.detach())
        net32.eval()
        with torch.no_grad():
            plain = net32(x, t)
        net32.train()
        scale = float(outs[0].abs().max())
        assert float((outs[0] - outs[1]).abs().max()) < 3e-2 * scale
        assert float((outs[0] - plain).abs().max()) > 4 * float((outs[0] - outs[1]).abs().max())      # (the dropout is visible: the check has power)
        ref = dict(net32.named_parameters())
        for (k, a), (_, b) in zip(net32.named_parameters(), net16.named_parameters()):
            sc_ = ref["out_net.0.pool.0.weight"].grad.norm() if k == "out_net.0.pool.0.bias" else a.grad.norm()
            rel_ = float((a.grad - b.grad).norm() / (sc_ + 1e-30))
            assert rel_ < 5e-2 and torch.isfinite(b.grad).all(), (Bn, P, k, rel_)


@pytest.mark.gpu
def test_layernorm_folded_inference_equals_the_plain_sequence(golden):
    """At large token counts the bf16 inference forward folds every LayerNorm into the products around it (row scale + rank-one
    correction in the consumer's epilogue, statistics from the producer's, the residual's LayerNorm recomputed where it is added):
    same network, same clouds as the plain kernel sequence -- which a forward that keeps a stash still runs -- and as the exact
    fp32 form, to bf16 accuracy; the prepared weight image and the per-call one agree bit for bit."""
    from so3x import backend as B
    net16, _ = full_net(golden, "bf16")
    net32, _ = full_net(golden, "fp32")
    net16, net32 = net16.to(DEV).eval(), net32.to(DEV).eval()
    gen = torch.Generator().manual_seed(17)
    Bn, P = 24, 2048                               # 49152 tokens: every product on the persistent 256-wide kernel
    x = (torch.randn(Bn, P, 3, generator=gen) * 0.5).to(DEV)
    t = torch.randint(0, 1000, (Bn,), generator=gen).to(DEV)
    flat = net16.flat_data()
    with torch.no_grad():
        folded = net16(x, t)                                                     # prepared image, no stash: the folded path
        folded_unprepared = B.planenet_fwd(flat, x, t, *net16.cfg)[0]             # the image built inside the call
        plain = B.planenet_fwd(flat, x, t, *net16.cfg, want_stash=True)[0]        # a stash: the plain sequence
        exact = net32(x, t)
    assert torch.equal(folded, folded_unprepared)
    scale = float(exact.abs().max())
    assert float((folded - plain).abs().max()) < 2e-2 * scale and not torch.equal(folded, plain)
    assert float((folded - exact).abs().max()) < 3e-2 * scale
    assert float((plain - exact).abs().max()) < 3e-2 * scale


@pytest.mark.gpu
def test_bf16_training_with_dropout_learns(golden):
    """end to end at the aircraft task's width: training mode, dropout 0.1, bf16 kernels, torch.optim.Adam on the flat gradient -- a
    fixed regression batch is fitted (finite losses, down by more than half in 25 steps)"""
    from so3x import rng
    from so3x.models import PlaneNet
    torch.manual_seed(4)
    net = PlaneNet(precision="bf16").to(DEV).train()          # torch's default dropout 0.1, as the reference builds it
    rng.manual_seed(4)
    x = torch.randn(8, 64, 3, device=DEV) * 0.5
    t = torch.randint(0, 1000, (8,), device=DEV)
    target = torch.randn(8, 3, device=DEV)
    opt = torch.optim.Adam(net.parameters(), lr=2e-4)
    losses = []
    for _ in range(25):
        opt.zero_grad(set_to_none=True)
        loss = (net(x, t) - target).square().mean()
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    assert all(np.isfinite(losses)) and min(losses[-3:]) < 0.5 * losses[0], losses
