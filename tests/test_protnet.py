"""ProtNet on the hand-written kernels (reference models.py:212-319, prot_train.py:75-104):
so3x_protnet_fwd / so3x_protnet_bwd against values the reference's own models.ProtNet produced on ragged synthetic complexes
(tools/make_golden.py protnet -> tests/golden/protnet.npz).

Two configurations: `small` (dim 32, 2 heads, t_depth 2, c_depth 4; chains of 1 .. 24 residues) and `default` (the class defaults
dim 64, 4 heads, t_depth 4, c_depth 3; chains of 40 .. 256 residues).  The weights (0.8 M / 2.3 M parameters) are rebuilt here from
the same seeds the fixture was made with; float64 checksums in the fixture pin them."""
from collections import namedtuple

import numpy as np
import pytest
import torch

DEV = "cuda:0"
ProtData = namedtuple("ProtData", ["residues", "positions", "angles"])


def protnet_perturb(net, seed):
    """tools/make_golden.py:protnet_perturb, verbatim"""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for _, p in sorted(net.named_parameters()):
            p.add_(torch.randn(p.shape, generator=g) * (0.05 * float(p.abs().mean()) + 1e-3))


def synthetic_complexes(lengths, seed):
    """tools/make_golden.py:synthetic_complexes, verbatim (ProtData is this module's namedtuple)"""
    g = torch.Generator().manual_seed(seed)

    def chain(L, centre):
        res = torch.zeros(L, 21)
        res[torch.arange(L), torch.randint(0, 21, (L,), generator=g)] = 1.0
        pos = torch.randn(L, 3, generator=g) * 8.0 + centre
        v1 = torch.nn.functional.normalize(torch.randn(L, 3, generator=g), dim=-1)
        v2 = torch.nn.functional.normalize(torch.randn(L, 3, generator=g), dim=-1)
        return ProtData(res, pos, torch.stack((v1, v2, torch.cross(v1, v2, dim=-1)), dim=1))
    out = []
    for lr, ll in lengths:
        c = torch.randn(3, generator=g) * 5.0
        out.append((chain(lr, c), chain(ll, c + 12.0)))
    return out


def build(golden, tag, precision="fp32"):
    """(net on the CPU with the fixture's weights, the fixture's complexes, the fixture)"""
    from so3x.models import ProtNet
    g = golden["protnet"]
    dim, heads, t_depth, c_depth = (int(v) for v in g[tag + "_cfg"])
    torch.manual_seed(31)
    net = ProtNet(dim=dim, heads=heads, t_depth=t_depth, c_depth=c_depth, precision=precision, dropout=0.0).eval()
    protnet_perturb(net, 7)
    for k, v in net.state_dict().items():                     # the weights ARE the ones the fixture was made with
        chk = g[f"{tag}_chk_{k}"]
        v64 = v.double()
        assert abs(float(v64.sum()) - chk[0]) <= 1e-9 * max(1.0, abs(chk[0])) and abs(float(v64.norm()) - chk[1]) <= 1e-9 * max(chk[1], 1e-30), k
    lengths = [tuple(int(x) for x in row) for row in g[tag + "_lengths"]]
    return net, synthetic_complexes(lengths, 101 if tag == "small" else 202), g


def to_dev(data):
    return [tuple(ProtData(*(a.to(DEV) for a in c)) for c in pair) for pair in data]


# ------------------------------------------------------------------------------------------------ CPU: layout and host logic
@pytest.mark.parametrize("tag", ["small", "default"])
def test_module_mirrors_the_reference_constructor_and_layout(golden, tag):
    """same seed -> the reference's initial weights (the checksums inside build()); the parameters sit in ONE flat buffer in
    state_dict order and the C library agrees about the count; the torch cross-check path reproduces the reference's outputs"""
    from so3x import backend as B
    net, data, g = build(golden, tag)
    flat = net.flat_data()
    assert flat.numel() == sum(p.numel() for p in net.parameters()) == B.protnet_param_count(*net.cfg)
    off = 0
    for k, v in net.state_dict().items():
        assert torch.equal(flat[off:off + v.numel()].view(v.shape), v), k
        off += v.numel()
    keys = list(net.state_dict().keys())
    assert keys[0] == "pos_emb.positional.weight" and keys[-1] == "last.4.bias"
    assert any(k.startswith("lig_tf.encoder.layers.0.self_attn.in_proj_weight") for k in keys) and "rec_tf.encoder.norm.weight" in keys
    with torch.no_grad():
        out = net.forward_torch(data, torch.from_numpy(g[tag + "_t"]))
    full = torch.cat((out.rot_g, out.shift_g), -1).numpy()
    assert np.abs(full - g[tag + "_out"]).max() < 2e-5


def test_protbatch_is_the_concatenation_with_offsets(golden):
    from so3x import backend as B
    _, data, _ = build(golden, "small")
    # (host-only check of the CSR layout: runs without a GPU by bypassing the device check)
    lens = [(r.positions.shape[0], l.positions.shape[0]) for r, l in data]
    rec_off = np.concatenate(([0], np.cumsum([n[0] for n in lens])))
    assert rec_off[-1] == sum(n[0] for n in lens) and max(max(n) for n in lens) == 24
    assert hasattr(B, "ProtBatch") and hasattr(B.ProtBatch, "from_pairs") and hasattr(B.ProtBatch, "with_ligands")


# ------------------------------------------------------------------------------------------------ GPU parity
def valid_rows(enc, lengths, max_len):
    """[2 B, max_len, d] padded layout -> (receptor rows, ligand rows), valid residues only, concatenated over the complexes"""
    B = len(lengths)
    rec = torch.cat([enc[i, :lengths[i][0]] for i in range(B)])
    lig = torch.cat([enc[B + i, :lengths[i][1]] for i in range(B)])
    return rec, lig


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["small", "default"])
def test_fp32_forward_vs_reference(golden, tag):
    """output, the head's 198-wide input and rec_tf's output at every valid residue of both chains: 1e-5 against the reference's
    float32 run, and no further from its float64 run than the reference's own float32 run is (x 3)"""
    net, data, g = build(golden, tag)
    net = net.to(DEV)
    t = torch.from_numpy(g[tag + "_t"]).to(DEV)
    lengths = [tuple(int(x) for x in row) for row in g[tag + "_lengths"]]
    with torch.no_grad():
        out, pool, enc = net(to_dev(data), t, want_internals=True)
        plain = net(to_dev(data), t)
    assert torch.equal(torch.cat((plain.rot_g, plain.shift_g), -1), out) and plain.rot_g.shape == (len(lengths), 3)
    out, pool = out.cpu().numpy(), pool.cpu().numpy()
    rec, lig = valid_rows(enc.cpu(), lengths, max(max(n) for n in lengths))
    for name, got, ref32, ref64 in (("out", out, g[tag + "_out"], g[tag + "_out64"]), ("pool", pool, g[tag + "_pool"], g[tag + "_pool64"]),
                                    ("rec_tf_out", rec.numpy(), g[tag + "_rec_tf_out"], g[tag + "_rec_tf_out64"]),
                                    ("lig_tf_out", lig.numpy(), g[tag + "_lig_tf_out"], g[tag + "_lig_tf_out64"])):
        scale = max(1.0, float(np.abs(ref64).max()))
        err32 = float(np.abs(got - ref32).max()) / scale
        err64 = float(np.abs(got - ref64).max()) / scale
        ref_err = float(np.abs(ref32 - ref64).max()) / scale
        assert err32 < 1e-5, (name, err32)
        assert err64 < max(3 * ref_err, 2e-6), (name, err64, ref_err)


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["small", "default"])
def test_fp32_backward_vs_reference_autograd(golden, tag):
    """every parameter's gradient of sum(out * dout) against the reference's autograd: per tensor the L2 norm and the sampled (or
    full) entries within 2e-4 of the tensor's scale (measured ~1e-5; the reference's own float32 gradients sit ~1e-5 from its
    float64 ones); lig_tf, which the reference never runs, gets zeros where the reference leaves None"""
    net, data, g = build(golden, tag)
    net = net.to(DEV).train()        # (training mode: the kernels run the eval arithmetic either way, see the class docstring)
    t = torch.from_numpy(g[tag + "_t"]).to(DEV)
    dout = torch.from_numpy(g[tag + "_dout"]).to(DEV)
    out = net(to_dev(data), t)
    full = torch.cat((out.rot_g, out.shift_g), -1)
    assert float((full.detach().cpu() - torch.from_numpy(g[tag + "_out"])).abs().max()) < 1e-5
    (full * dout).sum().backward()
    worst = 0.0
    for k, p in net.named_parameters():
        grad = p.grad.detach().cpu().reshape(-1)
        if f"{tag}_gnone_{k}" in g.files:
            assert k.startswith("lig_tf.") and float(grad.abs().max()) == 0.0, k
            continue
        gsum, pick64 = g[f"{tag}_gsum_{k}"], g[f"{tag}_gpick64_{k}"]
        stride = 1 if grad.numel() <= 4096 else max(1, grad.numel() // 64)
        got = grad[::stride][:4096 if stride == 1 else 64].double().numpy()
        scale = max(float(gsum[2]) / np.sqrt(grad.numel()), 1e-12)       # the tensor's RMS entry (float64 reference)
        err = float(np.abs(got - pick64).max()) / scale
        nerr = abs(float(grad.double().norm()) - float(gsum[2])) / max(float(gsum[2]), 1e-12)
        worst = max(worst, err * 1e-2, nerr)
        assert nerr < 2e-4, (k, "norm", nerr)
        assert err < 2e-2, (k, "entries", err)      # entries relative to the RMS entry: small entries of a tensor carry absolute, not relative, error
    assert worst < 2e-4


@pytest.mark.gpu
def test_ragged_batch_equals_its_complexes_one_by_one(golden):
    """padding is invisible: a batch of ragged complexes gives, complex by complex, what each complex gives alone (max_len then is
    its own longer chain) -- outputs to float32 round-off, and the batch gradient is the sum of the single gradients"""
    net, data, g = build(golden, "small")
    net = net.to(DEV)
    t = torch.from_numpy(g["small_t"]).to(DEV)
    dout = torch.from_numpy(g["small_dout"]).to(DEV)
    dd = to_dev(data)
    out = net(dd, t)
    full = torch.cat((out.rot_g, out.shift_g), -1)
    net.zero_grad(set_to_none=True)
    (full * dout).sum().backward()
    gb = net.gather_flat_grad().clone()
    acc = torch.zeros_like(gb)
    for i in range(len(dd)):
        net.zero_grad(set_to_none=True)
        o = net(dd[i:i + 1], t[i:i + 1])
        fi = torch.cat((o.rot_g, o.shift_g), -1)
        assert float((fi - full[i:i + 1]).abs().max()) < 2e-6
        (fi * dout[i:i + 1]).sum().backward()
        acc += net.gather_flat_grad()
    assert float((acc - gb).abs().max()) < 2e-5 * max(1.0, float(gb.abs().max()))


@pytest.mark.gpu
def test_error_paths(golden):
    from so3x import backend as B
    net, data, g = build(golden, "small")
    net = net.to(DEV)
    with pytest.raises(Exception):
        net(to_dev(data), torch.zeros(2, dtype=torch.long, device=DEV))      # one timestep per complex
    with pytest.raises(Exception):
        net(data, torch.from_numpy(g["small_t"]).to(DEV))                    # CPU tensors are refused
    with pytest.raises(ValueError):
        B.protnet_param_count(dim=30)                                        # dim % 4


# ------------------------------------------------------------------------------------------------ the caller: ProjectedSE3Diffusion
@pytest.mark.gpu
def test_prot_projection_on_a_batch_equals_the_list_form(golden):
    """ProtProjection (prot_util.py:102-117) given a ProtBatch moves every ligand in one launch and hands ProtNet the same numbers
    the reference-style list of (receptor, moved ligand) pairs does"""
    from so3x import backend as B
    from so3x.se3 import AffineT, ProtProjection
    net, data, g = build(golden, "small")
    net = net.to(DEV)
    dd = to_dev(data)
    n = len(dd)
    gen = torch.Generator(device=DEV).manual_seed(3)
    tf = AffineT(B.quat_to_rmat(torch.randn(n, 4, device=DEV, generator=gen)), torch.randn(n, 3, device=DEV, generator=gen))
    as_list = ProtProjection(dd)(tf)
    as_batch = ProtProjection(B.ProtBatch.from_pairs(dd))(tf)
    assert isinstance(as_list, list) and isinstance(as_batch, B.ProtBatch)
    ref = B.ProtBatch.from_pairs(as_list)
    assert float((ref.lig[1] - as_batch.lig[1]).abs().max()) < 1e-5 and float((ref.lig[2] - as_batch.lig[2]).abs().max()) < 1e-6
    assert torch.equal(ref.rec[1], as_batch.rec[1]) and torch.equal(ref.lig_off, as_batch.lig_off)
    t = torch.from_numpy(g["small_t"]).to(DEV)
    with torch.no_grad():
        a, b = net(as_list, t), net(as_batch, t)
    assert float((a.rot_g - b.rot_g).abs().max()) < 1e-5 and float((a.shift_g - b.shift_g).abs().max()) < 1e-5


@pytest.mark.gpu
def test_projected_se3_diffusion_training_step_with_protnet(golden):
    """prot_train.py:78-108 on the kernels: diff_model = ProjectedSE3Diffusion(ProtNet); loss = diff_model(true_pos, projection);
    loss.backward(); optim.step().  (a) with explicit noise draws the loss and EVERY parameter gradient equal those of the same
    step with the network run by torch's own modules (forward_torch: the reference's arithmetic); (b) a few Adam steps on one
    batch bring the loss down; (c) one reverse step runs through the same projection."""
    from so3x import backend as B
    from so3x.se3 import AffineT, ProjectedSE3Diffusion, ProtProjection
    net, data, g = build(golden, "small")
    net = net.to(DEV).train()
    dd = to_dev(data)
    n = len(dd)
    proc = ProjectedSE3Diffusion(net, timesteps=100).to(DEV)
    true_pos = AffineT(torch.eye(3, device=DEV).expand(n, 3, 3).contiguous(), torch.zeros(n, 3, device=DEV))
    gen = torch.Generator(device=DEV).manual_seed(11)
    t = torch.randint(0, 100, (n,), device=DEV, generator=gen)
    axes = torch.randn(n, 3, device=DEV, generator=gen)
    unif = torch.rand(n, device=DEV, generator=gen)
    znorm = torch.randn(n, 3, device=DEV, generator=gen)
    proc.projection = ProtProjection(B.ProtBatch.from_pairs(dd))
    loss = proc.p_losses(true_pos, t, axes=axes, unif=unif, znorm=znorm)
    net.zero_grad(set_to_none=True)
    loss.backward()
    got = {k: p.grad.detach().clone() for k, p in net.named_parameters()}
    # the same step, network through torch's modules (eval mode: the kernels run the eval arithmetic)
    import copy
    ref_net = copy.deepcopy(net).eval()
    ref_net.zero_grad(set_to_none=True)

    class TorchNet(torch.nn.Module):
        def forward(self, x, tt):
            return ref_net.forward_torch(x, tt)
    ref = ProjectedSE3Diffusion(TorchNet(), timesteps=100).to(DEV)
    ref.projection = ProtProjection(dd)
    ref_loss = ref.p_losses(true_pos, t, axes=axes, unif=unif, znorm=znorm)
    ref_loss.backward()
    assert abs(float(loss) - float(ref_loss)) < 2e-5 * max(1.0, abs(float(ref_loss)))
    for k, p in ref_net.named_parameters():
        if p.grad is None:
            assert k.startswith("lig_tf.") and float(got[k].abs().max()) == 0.0
            continue
        scale = max(float(p.grad.abs().max()), 1e-8)
        assert float((got[k] - p.grad).abs().max()) < 5e-4 * scale, k
    # (b)
    opt = torch.optim.Adam(net.parameters(), lr=2e-3)
    losses = []
    for _ in range(30):
        opt.zero_grad(set_to_none=True)
        l_ = proc.p_losses(true_pos, t, axes=axes, unif=unif, znorm=znorm)
        l_.backward()
        opt.step()
        losses.append(float(l_))
    assert losses[-1] < 0.5 * losses[0], losses[::6]
    # (c)
    with torch.no_grad():
        x = proc.p_sample(AffineT(B.quat_to_rmat(torch.randn(n, 4, device=DEV)), torch.randn(n, 3, device=DEV)), torch.full((n,), 50, device=DEV, dtype=torch.long))
    assert torch.isfinite(x.rot).all() and torch.isfinite(x.shift).all()
    full = proc(true_pos, ProtProjection(dd))        # the reference's own call form (prot_train.py:104)
    assert torch.isfinite(full)


# ------------------------------------------------------------------------------------------------ bf16 form (inference)
@pytest.mark.gpu
def test_bf16_forward_vs_reference(golden):
    """the class-default width on the bf16 matrix-core kernels (compact ragged token stream, one workgroup per chain for the
    attention block, token-parallel fused feed-forward): output and the head's 198-wide input within 3e-2 of the reference's float64
    run, relative to the tensor's largest entry (measured ~5e-3); the exact-fp32 form on the same inputs sits 1e-5 away"""
    from so3x import backend as B
    net, data, g = build(golden, "default", precision="bf16")
    net = net.to(DEV)
    t = torch.from_numpy(g["default_t"]).to(DEV)
    batch = B.ProtBatch.from_pairs(to_dev(data))
    with torch.no_grad():
        out = net(batch, t)
        _, _, pool, _ = B.protnet_fwd(net.flat_params_nograd(), batch, t, *net.cfg, precision=B.PREC_BF16, want_pool=True)
    full = torch.cat((out.rot_g, out.shift_g), -1).cpu().numpy()
    for name, got, ref in (("out", full, g["default_out64"]), ("pool", pool.cpu().numpy(), g["default_pool64"])):
        assert np.isfinite(got).all(), name
        err = float(np.abs(got - ref).max()) / max(1.0, float(np.abs(ref).max()))
        assert err < 3e-2, (name, err)
    # pooled positions are weighted means of fp32 inputs: tighter than the bf16 features
    d = 64
    for sl in (slice(2 * d, 2 * d + 3), slice(3 * d + 3, 3 * d + 6)):
        ref = g["default_pool64"][:, sl]
        assert float(np.abs(pool.cpu().numpy()[:, sl] - ref).max()) < 2e-2 * max(1.0, float(np.abs(ref).max()))


@pytest.mark.gpu
def test_bf16_ragged_batch_equals_its_complexes_one_by_one(golden):
    """the compact stream has no padded rows: a batch gives complex by complex what each complex gives alone, BIT FOR BIT (a chain's
    arithmetic does not depend on its neighbours: same kernels, same order of operations per chain)"""
    from so3x import backend as B
    net, data, g = build(golden, "default", precision="bf16")
    net = net.to(DEV)
    t = torch.from_numpy(g["default_t"]).to(DEV)
    dd = to_dev(data)
    with torch.no_grad():
        full = net(dd, t)
        for i in range(len(dd)):
            one = net(dd[i:i + 1], t[i:i + 1])
            assert torch.equal(one.rot_g, full.rot_g[i:i + 1]) and torch.equal(one.shift_g, full.shift_g[i:i + 1]), i
    with pytest.raises(Exception):       # the bf16 form is the class-default width only
        from so3x.models import ProtNet
        small = ProtNet(dim=32, heads=2, t_depth=2, c_depth=4, precision="bf16").to(DEV)
        with torch.no_grad():
            small(to_dev(build(golden, "small")[1]), torch.zeros(5, dtype=torch.long, device=DEV))


# ------------------------------------------------------------------------------------------------ training-mode dropout
def protnet_with_masks(net, data, t, masks, keep, Lp):
    """ProtNet's forward (reference models.py:275-319 over nn.TransformerEncoderLayer's post-norm arithmetic) in the kernels' own
    padded layout -- both chain kinds padded to Lp and stacked [2 B, Lp, .], receptors first -- with GIVEN dropout masks:
    masks[l] = (attention [2 B, H, Lp, Lp], block 1 [2 B, Lp, d], feed-forward [2 B, Lp, ffn], block 2 [2 B, Lp, d]).  Any dtype /
    device of `net`.  Test infrastructure."""
    import torch.nn.functional as Fn
    from torch.nn.utils.rnn import pad_sequence
    B = len(data)

    def padded(seqs):
        out = pad_sequence(seqs, batch_first=True)
        return Fn.pad(out, (0, 0) * (out.dim() - 2) + (0, Lp - out.shape[1]))
    chains = [p[0] for p in data] + [p[1] for p in data]
    pos = padded([c.positions for c in chains])
    ang = padded([c.angles for c in chains]).flatten(-2, -1)
    res = padded([net.res_conv(c.residues[None].transpose(-1, -2)).transpose(-1, -2)[0] for c in chains])
    msk = pos.any(dim=-1)
    h = torch.cat((res, net.pos_emb(pos), net.ang_emb(ang)), dim=-1)
    S, _, d = h.shape
    H = net.heads
    neg = torch.zeros(S, 1, 1, Lp, dtype=h.dtype).masked_fill(~msk[:, None, None, :], float("-inf"))
    for l, layer in enumerate(net.rec_tf.encoder.layers):
        ma, m1, mf, m2 = masks[l]
        qkv = Fn.linear(h, layer.self_attn.in_proj_weight, layer.self_attn.in_proj_bias)
        q, k, v = (z.reshape(S, Lp, H, d // H).transpose(1, 2) for z in qkv.split(d, dim=2))
        pr = torch.softmax(q @ k.transpose(2, 3) / (d // H) ** 0.5 + neg, dim=-1) * ma / keep
        y = layer.self_attn.out_proj((pr @ v).transpose(1, 2).reshape(S, Lp, d)) * m1 / keep
        x1 = layer.norm1(h + y)
        y2 = layer.linear2(torch.relu(layer.linear1(x1)) * mf / keep) * m2 / keep
        h = layer.norm2(x1 + y2)
    enc = net.rec_tf.encoder.norm(h)
    pools = [net.rec_emb_pool(enc[:B], msk[:B]), net.rec_pos_pool(enc[:B], pos[:B], msk[:B]),
             net.lig_emb_pool(enc[B:], msk[B:]), net.lig_pos_pool(enc[B:], pos[B:], msk[B:])]
    return net.last(torch.cat((net.time_emb(t), *pools), dim=-1))


@pytest.mark.gpu
def test_training_mode_dropout_vs_the_reference_arithmetic_with_the_same_masks(golden):
    """net.train() with nn.TransformerEncoderLayer's dropout (prot_train.py:75): the kernels' masks are a documented function of
    (seed, offset) (so3x.h; emulated here in numpy: tests/test_planenet.py:dropout_mask, whose Philox passes Random123's known
    answers).  torch's own mask stream cannot be matched, so parity is the reference's ARITHMETIC in float64 run with these masks:
    the output and every parameter gradient."""
    import copy
    from so3x import rng
    from test_planenet import dropout_mask
    net, data, g = build(golden, "small")
    net.dropout = 0.1
    net = net.to(DEV).train()
    t = torch.from_numpy(g["small_t"]).to(DEV)
    dout = torch.from_numpy(g["small_dout"]).to(DEV)
    rng.manual_seed(1234)
    seed, off = rng.seed(), rng.next_offset()
    rng.manual_seed(1234)                                   # the forward below draws the same (seed, offset)
    out = net(to_dev(data), t)
    full = torch.cat((out.rot_g, out.shift_g), -1)
    net.zero_grad(set_to_none=True)
    (full * dout).sum().backward()
    lengths = [tuple(int(x) for x in row) for row in g["small_lengths"]]
    S, Lp, d, H, F = 2 * len(lengths), max(max(n) for n in lengths), net.dim, net.heads, 2048
    keep = 1.0 - float(np.float32(0.1))
    masks = []
    for l in range(net.t_depth):
        m = [dropout_mask(S * H * Lp * Lp, 0.1, seed, off, l, 0).reshape(S, H, Lp, Lp), dropout_mask(S * Lp * d, 0.1, seed, off, l, 1).reshape(S, Lp, d),
             dropout_mask(S * Lp * F, 0.1, seed, off, l, 2).reshape(S, Lp, F), dropout_mask(S * Lp * d, 0.1, seed, off, l, 3).reshape(S, Lp, d)]
        masks.append([torch.from_numpy(a.astype(np.float64)) for a in m])
    ref = copy.deepcopy(net).cpu().double().eval()
    ref.zero_grad(set_to_none=True)
    data64 = [tuple(ProtData(*(a.double() for a in c)) for c in pair) for pair in data]
    want = protnet_with_masks(ref, data64, t.cpu(), masks, keep, Lp)
    (want * dout.cpu().double()).sum().backward()
    assert float((full.detach().cpu().double() - want.detach()).abs().max()) < 2e-5 * max(1.0, float(want.abs().max()))
    different = protnet_with_masks(ref, data64, t.cpu(), [[torch.ones_like(a) for a in m] for m in masks], 1.0, Lp)
    assert float((want - different).abs().max()) > 1e-3            # the masks matter
    for (k, p), (_, pr) in zip(net.named_parameters(), ref.named_parameters()):
        if pr.grad is None:
            assert k.startswith("lig_tf.")
            continue
        scale = max(float(pr.grad.abs().max()), 1e-8)
        assert float((p.grad.detach().cpu().double() - pr.grad).abs().max()) < 5e-4 * scale, k
    # eval mode: no dropout, whatever the attribute says
    net.eval()
    with torch.no_grad():
        a, b = net(to_dev(data), t), net(to_dev(data), t)
    assert torch.equal(a.rot_g, b.rot_g) and float((torch.cat((a.rot_g, a.shift_g), -1).cpu() - torch.from_numpy(g["small_out"])).abs().max()) < 1e-5


@pytest.mark.gpu
def test_bf16_edge_lengths_vs_the_fp32_form():
    """chain lengths that hit every branch of the bf16 kernels -- 1 residue, one short of / exactly / one past a 16-, 32- and 64-row
    tile, the 256-residue maximum, receptor shorter than its ligand -- against the exact-fp32 form on the same weights and inputs
    (itself pinned to the reference).  Gate: 5e-2 of the output's scale.  What bf16 operands cost is a property of the WEIGHTS, not of
    a length: every product rounds its operands to 8 bits, ~0.5 % of the activations' scale per encoder layer (measured per layer and
    per chain, tools/ab/protnet_edge_dbg.py; an fp32 residual stream and exact sines were tried and change nothing): the golden
    weights land at 2.2 % on the output, these at 3.4 %, the worst single complex below at 4.5 %."""
    from so3x.models import ProtNet
    lengths = [(1, 1), (15, 16), (17, 31), (32, 33), (63, 64), (65, 127), (128, 129), (255, 256), (256, 2), (3, 200)]
    data = to_dev(synthetic_complexes(lengths, 77))
    torch.manual_seed(5)
    net = ProtNet(precision="bf16", dropout=0.0).eval()
    protnet_perturb(net, 9)
    net = net.to(DEV)
    t = torch.randint(0, 1000, (len(lengths),), device=DEV)
    with torch.no_grad():
        got = net(data, t)
        net.precision = "fp32"
        want = net(data, t)
    a, b = torch.cat((got.rot_g, got.shift_g), -1), torch.cat((want.rot_g, want.shift_g), -1)
    assert torch.isfinite(a).all()
    assert float((a - b).abs().max()) < 5e-2 * max(1.0, float(b.abs().max())), float((a - b).abs().max())
    # rec_tf's output, residue by residue (padded layout): no residue further than a fifth of the encoder's scale, the mean error a
    # few per cent of it -- a wrong mask or tile boundary shows as O(1) differences at the affected residues
    from so3x import backend as B
    batch = B.ProtBatch.from_pairs(data)
    with torch.no_grad():
        e16 = B.protnet_fwd(net.flat_params_nograd(), batch, t, *net.cfg, precision=B.PREC_BF16, want_encoding=True)[3]
        e32 = B.protnet_fwd(net.flat_params_nograd(), batch, t, *net.cfg, precision=B.PREC_F32, want_encoding=True)[3]
    scale = float(e32.abs().max())
    for i, (lr, ll) in enumerate(lengths):
        for s_, L in ((i, lr), (len(lengths) + i, ll)):
            d = (e16[s_, :L] - e32[s_, :L]).abs()
            assert float(d.max()) < 0.25 * scale and float(d.mean()) < 0.05 * scale, (s_, L, float(d.max()), float(d.mean()), scale)
            assert float(e16[s_, L:].abs().max() if L < e16.shape[1] else 0.0) == 0.0
    # longer than the bf16 kernels hold in LDS: refused, not truncated
    long = to_dev(synthetic_complexes([(257, 10)], 78))
    net.precision = "bf16"
    with pytest.raises(Exception):
        with torch.no_grad():
            net(long, t[:1])
    net.precision = "fp32"
    with torch.no_grad():
        assert torch.isfinite(net(long, t[:1]).rot_g).all()
