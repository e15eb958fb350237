"""The wide residual score network and training script of the reference's so3_lock_train.py:
RotPredict(d_model=255) = six ResLayer(Linear(255,255)+SiLU) blocks and a Linear(255,3) head on
[R(9), sin(123), cos(123)], same constructor, forward signature and state_dict keys
(reference so3_lock_train.py:11-59), evaluated by the streaming-weight MFMA kernels of libso3x
(so3x_resnet_fwd; fused into the reverse chain by SO3Diffusion -> so3x_resnet_p_sample_chain)."""
import argparse
import json
import os
import time
from math import pi

import torch
from torch import nn

from . import backend as _b
from .flat import FlatParamsMixin
from .models import SinusoidalPosEmb, ResLayer

__all__ = ["RotPredict", "BATCH", "main"]

BATCH = 32
_PRECISIONS = {"fp32": _b.PREC_F32, "bf16": _b.PREC_BF16}


class _ResNetFn(torch.autograd.Function):
    """autograd bridge: only the 392,448 parameters need gradients (loss_type='skewvec': inputs and targets carry none)"""

    @staticmethod
    def forward(ctx, x, t, flat_params, precision, t_table):
        ctx.precision = precision
        ctx.t_table = t_table
        out, stash = _b.resnet_fwd_stash(flat_params, x, t, t_table, precision)  # layer inputs + pre-activations for the backward
        ctx.save_for_backward(x, t, flat_params, stash)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, t, flat_params, stash = ctx.saved_tensors
        dparams = _b.resnet_bwd(flat_params, x, t, dout.contiguous(), ctx.t_table, ctx.precision, stash=stash)
        return None, None, dparams, None, None


class RotPredict(FlatParamsMixin, nn.Module):
    kind = "resnet255"  # which fused kernels SO3Diffusion dispatches to

    def __init__(self, d_model=255, out_type="rotmat", in_type="rotmat", precision="fp32"):
        super().__init__()
        self.in_type = in_type
        self.out_type = out_type
        if in_type != "rotmat" or d_model != 255:
            raise NotImplementedError("so3x: the wide score network is built for in_type='rotmat', d_model=255")
        if out_type not in ("skewvec", "rotmat"):
            raise ValueError(f"Unexpected out_type: {out_type}")  # the reference builds this error without raising it (so3_lock_train.py:24)
        if precision not in _PRECISIONS:
            raise ValueError(f"precision must be one of {list(_PRECISIONS)}")
        self.precision = precision
        self.d_out = 3 if out_type == "skewvec" else 6  # "rotmat": six2rmat of a 6-wide head (so3_lock_train.py:19-22, 57-58)
        self.time_embedding = SinusoidalPosEmb(d_model - 9)
        self.net = nn.Sequential(*[ResLayer(nn.Sequential(nn.Linear(d_model, d_model), nn.SiLU())) for _ in range(6)],
                                 nn.Linear(d_model, self.d_out))
        # the kernels gather per-timestep input rows from a [T][256] table: T must bound every t.  SO3Diffusion passes
        # its num_timesteps per call; this attribute is the default for direct calls.
        self.t_table = 1000
        # the 392,448 (skewvec) / 393,216 (rotmat) parameters live in ONE flat buffer in state_dict order (so3x.flat)
        self._init_flat()

    @property
    def precision_code(self) -> int:
        return _PRECISIONS[self.precision]

    def forward(self, x: torch.Tensor, t: torch.Tensor, t_table: int = None, raw: bool = False):
        """raw=True returns the network's [.., d_out] outputs without the six2rmat of out_type="rotmat" (SO3Diffusion's
        prevstep loss applies it inside its own fused kernel)"""
        tt = self.t_table if t_table is None else int(t_table)
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.net.parameters()):
            out = _ResNetFn.apply(x, t, self.flat_params(), self.precision_code, tt)
        else:
            out = _b.resnet_fwd(self.flat_params_nograd(), x, t, tt, self.precision_code)
        return _b.six2rmat(out) if (self.out_type == "rotmat" and not raw) else out


def main(argv=None):
    """Training loop of the reference's so3_lock_train.py:64-97 (data = the so3_lerp arc between two Euler rotations,
    Adam 3e-4), data-parallel over the GPUs of one node when launched with torchrun."""
    from .diffusion import SO3Diffusion
    from .util import euler_to_rmat, so3_lerp
    from . import parallel
    from . import optim as so3x_optim

    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=BATCH, help="global batch (reference: 32)")
    ap.add_argument("--steps", type=int, default=100000)
    ap.add_argument("--timesteps", type=int, default=1000)
    ap.add_argument("--precision", default="fp32", choices=list(_PRECISIONS))
    ap.add_argument("--lr", type=float, default=3e-4)
    ap.add_argument("--log-every", type=int, default=10)
    ap.add_argument("--save-every", type=int, default=1000)
    ap.add_argument("--graph", action="store_true", help="replay the whole step as a captured hipGraph (so3x.graphs.TrainStepGraph)")
    ap.add_argument("--optimizer", default="so3x", choices=["so3x", "torch"])
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--weights", default="weights/weights_so3_lock.pt")
    args = ap.parse_args(argv)

    ctx = parallel.init()
    device = ctx.device
    torch.manual_seed(args.seed)
    net = RotPredict(out_type="skewvec", precision=args.precision).to(device)
    net.train()
    parallel.broadcast_parameters(net, ctx)
    torch.cuda.manual_seed(args.seed + 7919 * (ctx.rank + 1))  # per-rank timesteps (diffusion.py:373 draws them on the device)
    process = SO3Diffusion(net, timesteps=args.timesteps, loss_type="skewvec").to(device)
    if args.optimizer == "so3x":
        optim = so3x_optim.Adam(net, lr=args.lr)
    else:
        optim = torch.optim.Adam(net.parameters(), lr=args.lr, fused=True, capturable=args.graph)
    R_1 = euler_to_rmat(torch.tensor(0.0), torch.tensor(pi / 3), torch.tensor(0.0))[None].to(device)
    R_2 = euler_to_rmat(torch.tensor(0.0), torch.tensor(2 * pi / 3), torch.tensor(0.0))[None].to(device)
    lo, hi = parallel.shard_range(args.batch, ctx.rank, ctx.world_size)
    process.index_base = lo
    gen = torch.Generator(device=device).manual_seed(1234 + ctx.rank)
    t0 = time.time()
    sumloss = 0.0
    graph = None
    if args.graph:
        from .graphs import TrainStepGraph
        graph = TrainStepGraph(process, optim, (hi - lo, 3, 3), ctx=ctx, n_global=args.batch)
    for i in range(1, args.steps + 1):
        weight = torch.rand(hi - lo, 1, device=device, generator=gen)
        truepos = so3_lerp(R_1, R_2, weight)
        if graph is not None:
            loss = graph.step(truepos)  # (the reference's skip-on-NaN needs the loss on the host before the update: eager mode only)
        else:
            loss = process(truepos)
            # so3_lock_train.py:83-84 skips the update on a NaN loss.  The decision is made for ALL ranks together (a rank
            # that skipped alone would miss the collectives the others enter and pair its next all-reduce with theirs)
            if parallel.any_rank_true(torch.isnan(loss).any(), ctx):
                continue
            optim.zero_grad()
            loss.backward()
            parallel.allreduce_gradients(net, ctx, n_local=hi - lo, n_global=args.batch, optimizer=optim)
            optim.step()
        sumloss += parallel.mean_scalar(loss.detach(), ctx)
        if i % args.log_every == 0:
            if ctx.rank == 0:
                print(json.dumps({"step": i, "loss": sumloss / args.log_every, "elapsed_s": round(time.time() - t0, 3)}), flush=True)
            sumloss = 0.0
        if i % args.save_every == 0 and ctx.rank == 0:
            os.makedirs(os.path.dirname(args.weights) or ".", exist_ok=True)
            torch.save(net.state_dict(), args.weights)
    parallel.finalize(ctx)
    return net


if __name__ == "__main__":
    main()
