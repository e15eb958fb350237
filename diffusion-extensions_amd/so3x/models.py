"""SinusoidalPosEmb with the reference's interface (reference models.py:13-25).
On the hot path the embedding is evaluated inside the fused score-network kernels;
this module exists for API/state_dict compatibility and standalone use."""
import math

import torch
from torch import nn

from .flat import FlatParamsMixin

__all__ = ["SinusoidalPosEmb", "Siren", "ResLayer", "PointCloudProj", "PoolRN", "PoolPos", "TransformerEnc2", "PlaneNet", "ProtNet", "RES_COUNT"]

RES_COUNT = 21   # prot_util.py:9-35: 20 residue types + "---" (unknown / padding)


class SinusoidalPosEmb(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.dim = dim

    def forward(self, x):
        half = self.dim // 2
        freqs = torch.exp(torch.arange(half, device=x.device) * -(math.log(10000) / (half - 1)))
        ang = x[:, None] * freqs[None, :]
        return torch.cat((ang.sin(), ang.cos()), dim=-1)


class Siren(nn.Module):
    """Sine-activated positional encoding layer (reference models.py:37-72; imported by so3_train.py:6 and unused on the
    SO(3) path): sin(Linear(x)) with the SIREN initialisation, optionally followed by a Linear.  Plain torch."""

    def __init__(self, in_channels, out_channels, scale=1, optimize=True, post_scale=True):
        super().__init__()
        self.positional = nn.Linear(in_features=in_channels, out_features=out_channels)
        bound = (6 / in_channels) ** 0.5
        nn.init.uniform_(self.positional.weight, -bound, bound)
        self.positional.weight.data *= scale
        nn.init.uniform_(self.positional.bias, -3.14159, 3.14159)  # biases cover +-pi
        self.post_scale = nn.Linear(out_channels, out_channels) if post_scale else None
        for param in self.parameters():
            param.requires_grad = optimize

    def forward(self, x):
        res = torch.sin(self.positional(x))
        return self.post_scale(res) if self.post_scale is not None else res


class ResLayer(nn.Module):
    """x + layer(x) (reference models.py:28-34); the container the wide score network's state_dict keys go through
    (`net.<i>.layer.0.weight`).  so3_lock_train.RotPredict evaluates the whole stack in the fused kernels."""

    def __init__(self, layer: nn.Module):
        super().__init__()
        self.layer = layer

    def forward(self, x):
        return x + self.layer(x)


class PointCloudProj(nn.Module):
    """data @ R^T for a batch of rotations R (reference models.py:75-91): the `projection` callable handed to
    ProjectedSO3Diffusion / ProjectedSE3Diffusion.  so3=False (Euler-angle input of the Euclidean baselines) converts with
    util.euler_to_rmat first."""

    def __init__(self, data, so3=True):
        super().__init__()
        self.data = data
        self.so3 = so3

    def forward(self, x):
        from . import backend as _b
        if not self.so3:
            from .util import euler_to_rmat
            x = euler_to_rmat(*torch.unbind(x, -1))
        return _b.rotate_cloud(x, self.data)


class PoolRN(nn.Module):
    """Learned-weight mean over the point axis (reference models.py:94-110): out = sum_p w_p lin(x_p) / sum_p w_p with
    w_p = sigmoid(Linear(x_p)) * mask_p.  mask: [.., P] booleans, None = all points.  (The reference's own default mask is
    built one axis too wide -- [.., P, 1] and then indexed [..., None], models.py:105-106 -- which only broadcasts when the
    batch equals the point count; the semantics here are those its masked callers get.)"""

    def __init__(self, dim):
        super().__init__()
        self.pool = nn.Sequential(nn.Linear(dim, 1), nn.Sigmoid())
        self.lin = nn.Linear(dim, dim)

    def forward(self, x, mask=None):
        if mask is None:
            mask = torch.ones(x.shape[:-1], dtype=torch.bool, device=x.device)
        weight = self.pool(x) * mask[..., None]
        w_sum = weight.sum(dim=-2, keepdim=True).clamp(min=1e-6)
        out = (self.lin(x) * weight).sum(dim=-2, keepdim=True) / w_sum
        return out[..., 0, :]


class PoolPos(nn.Module):
    """Learned-weight mean of the residue POSITIONS (reference models.py:113-127): out = sum_p w_p pos_p / sum_p w_p with
    w_p = sigmoid(Linear(x_p)) * mask_p.  mask: [.., P] booleans (see PoolRN for the reference's default-mask shape)."""

    def __init__(self, dim_pool):
        super().__init__()
        self.pool = nn.Sequential(nn.Linear(dim_pool, 1), nn.Sigmoid())

    def forward(self, x, pos, mask=None):
        if mask is None:
            mask = torch.ones(x.shape[:-1], dtype=torch.bool, device=x.device)
        weight = self.pool(x) * mask[..., None]
        w_sum = weight.sum(dim=-2, keepdim=True).clamp(min=1e-6)
        return ((pos * weight).sum(dim=-2, keepdim=True) / w_sum)[..., 0, :]


class TransformerEnc2(nn.Module):
    """nn.TransformerEncoder with a final LayerNorm on batch-first input (reference models.py:167-182)"""

    def __init__(self, dim=512, heads=4, layers=4):
        super().__init__()
        self.encoder = nn.TransformerEncoder(nn.TransformerEncoderLayer(dim, heads), layers, norm=nn.LayerNorm(dim, eps=1e-5))

    def forward(self, x, src_key_padding_mask=None):
        return self.encoder(x.transpose(0, 1), src_key_padding_mask=src_key_padding_mask).transpose(0, 1)


class _PlaneNetFn(torch.autograd.Function):
    """autograd bridge of the PlaneNet kernels: only the parameters carry gradients (the point clouds are projections of the
    noised pose, which needs none: reference diffusion.py:389-392)"""

    @staticmethod
    def forward(ctx, x, t, flat_params, cfg, drop):
        from . import backend as _b
        out, stash, _ = _b.planenet_fwd(flat_params, x, t, *cfg, want_stash=True, dropout_p=drop[0], seed=drop[1], rng_offset=drop[2])
        ctx.cfg, ctx.drop = cfg, drop
        ctx.save_for_backward(x, t, flat_params, stash)
        return out

    @staticmethod
    def backward(ctx, dout):
        from . import backend as _b
        x, t, flat_params, stash = ctx.saved_tensors
        p, seed, off = ctx.drop
        return None, None, _b.planenet_bwd(flat_params, x, t, dout.contiguous(), stash, *ctx.cfg, dropout_p=p, seed=seed, rng_offset=off), None, None


class PlaneNet(FlatParamsMixin, nn.Module):
    """The point-cloud pose denoiser of the aircraft task (reference models.py:185-210): SIREN position encoding of every
    point (dim/2) next to the timestep's sinusoidal embedding (dim/2), a `layers`-deep nn.TransformerEncoder over the points,
    PoolRN over the points and a Linear to the 3 skew-vector components -- the `denoise_fn` of ProjectedSO3Diffusion with
    PointCloudProj as the projection (aircraft_rotate.py:64-106).

    The modules below exist for the reference's constructor order (same seed -> same initial weights) and its state_dict keys;
    `forward` does not run them.  It runs the hand-written kernels of libso3x (so3x_planenet_fwd / so3x_planenet_bwd: embedding,
    encoder layers with attention, pooling and head, forward and backward) on the flat parameter buffer the module parameters are
    views of.  precision "fp32": every product on the exact-fp32 matrix-core instruction, any width; "bf16": the aircraft task's
    shape (dim 512, 4 heads, points a multiple of 64) with bf16 operands and activations.  `dropout` (torch's default 0.1, which
    the reference trains with: aircraft_rotate.py:66) is applied in training mode at nn.TransformerEncoderLayer's four sites by
    the kernels of either precision, from counter-based masks keyed by (so3x.rng seed, a fresh offset per forward) that the
    backward regenerates -- the two precisions draw the SAME masks.
    `forward_torch` runs the torch modules instead -- an explicit cross-check for tests, never taken implicitly.

    Returns [B, 3]: one prediction per cloud.  (The reference's forward ends in `out[..., 0, :]` on that [B, 3] tensor,
    models.py:210, i.e. it hands back sample 0's row only, and its pooling fails for batch != points -- see PoolRN.)"""

    def __init__(self, dim=512, heads=4, layers=4, precision="fp32", dropout=0.1):
        super().__init__()
        if precision not in ("fp32", "bf16"):
            raise ValueError("precision must be 'fp32' or 'bf16'")
        self.encoder = nn.TransformerEncoder(nn.TransformerEncoderLayer(dim, heads, dropout=dropout), layers)
        self.position_siren = Siren(in_channels=3, out_channels=dim // 2, scale=30)
        self.time_embedding = SinusoidalPosEmb(dim // 2)
        self.out_net = nn.Sequential(PoolRN(dim), nn.Linear(dim, 3))
        self.dim, self.heads, self.layers, self.precision, self.dropout = dim, heads, layers, precision, dropout
        self.ffn = self.encoder.layers[0].linear1.out_features
        from .flat import PreparedCache
        self._prep = PreparedCache(every=64)   # the bf16 weight image: inference reuses it while the parameters do not change
        self._init_flat()

    def _flat_root(self):
        return self

    @property
    def cfg(self):
        from . import backend as _b
        return (self.dim, self.heads, self.layers, self.ffn, _b.PREC_BF16 if self.precision == "bf16" else _b.PREC_F32)

    def _prepared(self):
        """the bf16 image of the weight matrices for the current parameters (so3x_planenet_prepare), rebuilt when any parameter's
        tensor version, the out-of-band update epoch or the buffer changes, and when the buffer's fingerprint does (writes through
        `p.data`: so3x.flat.PreparedCache, as SO3Diffusion's prepared sampling state)"""
        from . import backend as _b
        from .flat import PARAM_EPOCH
        flat = self.flat_params_nograd()
        key = (flat.data_ptr(), tuple(p._version for p in self._flat_params), PARAM_EPOCH[0], self.precision, flat.device)
        return self._prep.get(key, flat, lambda: _b.planenet_prepare(flat, *self.cfg))

    def invalidate_prepared(self):
        """for callers that rewrite the parameters behind torch's back and want the next forward to see it unconditionally"""
        self._prep.invalidate()

    def train(self, mode=True):
        self._prep.check_next()      # train() / eval() usually bracket a weight update: the next inference compares fingerprints
        return super().train(mode)

    def forward(self, x, t, want_encoding=False):
        from . import backend as _b
        from . import rng as _rng
        drop = (0.0, 0, 0)
        if self.training and self.dropout > 0:
            drop = (float(self.dropout), _rng.seed(), _rng.next_offset())   # a fresh mask set per training-mode forward
        if want_encoding:
            out, _, enc = _b.planenet_fwd(self.flat_params_nograd(), x, t, *self.cfg, want_stash=drop[0] > 0, want_encoding=True,
                                          prepared=None if drop[0] > 0 else self._prepared(), dropout_p=drop[0], seed=drop[1], rng_offset=drop[2])
            return out, enc
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            return _PlaneNetFn.apply(x, t, self.flat_params(), self.cfg, drop)
        if drop[0] > 0:   # a training-mode forward nobody differentiates: the same masks, the stash dropped
            return _b.planenet_fwd(self.flat_params_nograd(), x, t, *self.cfg, want_stash=True, dropout_p=drop[0], seed=drop[1], rng_offset=drop[2])[0]
        return _b.planenet_fwd(self.flat_params_nograd(), x, t, *self.cfg, prepared=self._prepared())[0]

    def forward_torch(self, x, t):
        """the same network through torch's own modules (any device): test infrastructure"""
        x_emb = self.position_siren(x)                                         # [B, P, dim/2]
        t_emb = self.time_embedding(t)                                         # [B, dim/2]
        t_in = torch.cat((x_emb, t_emb[:, None, :].expand(x_emb.shape)), dim=2)
        encoding = self.encoder(t_in.transpose(0, 1)).transpose(0, 1)          # sequence-first inside, as the reference
        return self.out_net(encoding)


class _ProtNetFn(torch.autograd.Function):
    """autograd bridge of the ProtNet kernels: only the parameters carry gradients (the complexes are a projection of the noised
    pose, reference diffusion.py:558-559)"""

    @staticmethod
    def forward(ctx, t, flat_params, cfg, batch, drop):
        from . import backend as _b
        out, stash, _, _ = _b.protnet_fwd(flat_params, batch, t, *cfg, want_stash=True, dropout_p=drop[0], seed=drop[1], rng_offset=drop[2])
        ctx.cfg, ctx.max_len, ctx.drop = cfg, batch.max_len, drop
        ctx.save_for_backward(flat_params, stash)
        return out

    @staticmethod
    def backward(ctx, dout):
        from . import backend as _b
        flat_params, stash = ctx.saved_tensors
        p, seed, off = ctx.drop
        return None, _b.protnet_bwd(flat_params, dout.contiguous(), stash, ctx.max_len, *ctx.cfg, dropout_p=p, seed=seed, rng_offset=off), None, None, None


class ProtNet(FlatParamsMixin, nn.Module):
    """The docking denoiser of prot_train.py (reference models.py:212-319): per chain a residue-type convolution stack, SIREN
    encodings of the CA positions and residue frames, a t_depth-deep transformer encoder with key-padding masks, PoolRN / PoolPos
    over the residues; the two chains' pools and the timestep embedding go through a small residual MLP to (rot_g, shift_g) -- the
    `denoise_fn` of ProjectedSE3Diffusion with ProtProjection as the projection (prot_train.py:78-104).

    The modules below exist for the reference's constructor order (same seed -> same initial weights) and its state_dict keys;
    `forward` does not run them.  It runs the hand-written kernels of libso3x (so3x_protnet_fwd / so3x_protnet_bwd) on the flat
    parameter buffer the module parameters are views of.  As in the reference BOTH chains are encoded by `rec_tf` (models.py:288,
    302); `lig_tf` is constructed, saved and loaded, never run, and gets a zero gradient (the reference leaves it None).
    `dropout` (torch's default 0.1, which the reference trains with: prot_train.py:75 `net.train()`) is applied in training mode at
    rec_tf's four sites per layer by the exact-fp32 kernels, from counter-based masks keyed by (so3x.rng seed, a fresh offset per
    forward) that the backward regenerates (so3x.h); eval mode -- sampling, prot_test.py -- runs the plain arithmetic.
    precision "fp32": exact-fp32 matrix-core products, any width, forward and backward; "bf16": the class-default width (dim 64,
    4 heads) with bf16 operands, inference only -- a forward that needs gradients runs the fp32 form.

    forward(x, t): x = a sequence of (receptor, ligand) ProtData pairs, as ProtProjection returns them, or a
    so3x.backend.ProtBatch (the same data already concatenated: what a loader builds once per batch); t int64 [B].
    Returns AffineGrad(rot_g [B, 3], shift_g [B, 3]) (se3=True) or the [B, 6] tensor."""

    def __init__(self, dim=64, heads=4, t_depth=4, c_depth=3, se3=True, precision="fp32", dropout=0.1):
        super().__init__()
        if precision not in ("fp32", "bf16"):
            raise ValueError("precision must be 'fp32' or 'bf16'")
        self.dropout = dropout
        pos_dim, ang_dim = dim // 2, dim // 4
        res_dim = dim - (pos_dim + ang_dim)
        self.se3 = se3
        self.time_emb = SinusoidalPosEmb(dim)
        self.pos_emb = Siren(3, pos_dim, scale=0.1)
        self.ang_emb = Siren(9, ang_dim)

        def conv(cin, cout):
            return nn.Conv1d(in_channels=cin, out_channels=cout, kernel_size=(3,), padding=(1,), stride=(1,))
        self.res_conv = nn.Sequential(conv(RES_COUNT, dim), nn.SiLU(inplace=True),
                                      *[ResLayer(nn.Sequential(conv(dim, dim), nn.SiLU(inplace=True))) for _ in range(c_depth - 2)],
                                      conv(dim, res_dim))
        self.lig_tf = TransformerEnc2(dim=dim, layers=t_depth, heads=heads)
        self.lig_emb_pool = PoolRN(dim)
        self.lig_pos_pool = PoolPos(dim)
        self.rec_tf = TransformerEnc2(dim=dim, layers=t_depth, heads=heads)
        self.rec_emb_pool = PoolRN(dim)
        self.rec_pos_pool = PoolPos(dim)
        self.last = nn.Sequential(nn.Sequential(nn.Linear(3 * dim + 6, dim), nn.SiLU(inplace=True)),
                                  *[ResLayer(nn.Sequential(nn.Linear(dim, dim), nn.SiLU(inplace=True))) for _ in range(3)],
                                  nn.Linear(dim, 6))
        self.dim, self.heads, self.t_depth, self.c_depth, self.precision = dim, heads, t_depth, c_depth, precision
        self._init_flat()

    def _flat_root(self):
        return self

    @property
    def cfg(self):
        return (self.dim, self.heads, self.t_depth, self.c_depth)

    def _wrap(self, out):
        from .se3 import AffineGrad
        return AffineGrad(rot_g=out[..., :3], shift_g=out[..., 3:]) if self.se3 else out

    def forward(self, x, t, want_internals=False):
        from . import backend as _b
        batch = x if isinstance(x, _b.ProtBatch) else _b.ProtBatch.from_pairs(x)
        if want_internals:    # (out, the head's input [B, 3 dim + 6], rec_tf's output in the padded layout): parity tests
            out, _, pool, enc = _b.protnet_fwd(self.flat_params_nograd(), batch, t, *self.cfg, want_pool=True, want_encoding=True)
            return out, pool, enc
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            from . import rng as _rng
            drop = (0.0, 0, 0)
            if self.training and self.dropout > 0:
                drop = (float(self.dropout), _rng.seed(), _rng.next_offset())   # a fresh mask set per training-mode forward
            return self._wrap(_ProtNetFn.apply(t, self.flat_params(), self.cfg, batch, drop))
        prec = _b.PREC_BF16 if self.precision == "bf16" else _b.PREC_F32
        return self._wrap(_b.protnet_fwd(self.flat_params_nograd(), batch, t, *self.cfg, precision=prec)[0])

    def forward_torch(self, x, t):
        """the same network through torch's own modules (any device): test infrastructure.  Follows models.py:275-319, ligand
        through rec_tf included."""
        from torch.nn.utils.rnn import pad_sequence
        pools = []
        for k, (emb_pool, pos_pool) in enumerate(((self.rec_emb_pool, self.rec_pos_pool), (self.lig_emb_pool, self.lig_pos_pool))):
            chains = [pair[k] for pair in x]
            pos = pad_sequence([c.positions for c in chains], batch_first=True)
            ang = pad_sequence([c.angles for c in chains], batch_first=True).flatten(-2, -1)
            res = pad_sequence([self.res_conv(c.residues[None].transpose(-1, -2)).transpose(-1, -2)[0] for c in chains], batch_first=True)
            msk = pos.any(dim=-1)
            enc = self.rec_tf(torch.cat((res, self.pos_emb(pos), self.ang_emb(ang)), dim=-1), src_key_padding_mask=msk.logical_not())
            pools += [emb_pool(enc, msk), pos_pool(enc, pos, msk)]
        return self._wrap(self.last(torch.cat((self.time_emb(t), *pools), dim=-1)))
