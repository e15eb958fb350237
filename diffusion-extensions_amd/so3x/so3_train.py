"""RotPredict score network with the reference's constructor, forward signature and
state_dict keys (reference so3_train.py:11-49), plus the training entry point.

forward/backward run in the fused MFMA kernels (so3x_mlp_fwd / so3x_mlp_bwd): the
time embedding, the 5 linear layers and the 4 SiLUs are one launch each way."""
import argparse
import json
import os
import time

import torch
from torch import nn

from . import backend as _b
from .flat import FlatParamsMixin
from .models import SinusoidalPosEmb

__all__ = ["RotPredict", "BATCH", "main"]

BATCH = 64
_PRECISIONS = {"fp32": _b.PREC_F32, "bf16": _b.PREC_BF16}


class _ScoreMLPFn(torch.autograd.Function):
    """autograd bridge: only the 17,358 parameters need gradients -- the rotation inputs
    and targets of loss_type='skewvec' carry none."""

    @staticmethod
    def forward(ctx, x, t, flat_params, precision, t_table):
        ctx.precision = precision
        ctx.t_table = t_table
        if precision == _b.PREC_BF16 and t_table > 0:
            # the fused backward's half: park the pre-activations (544 B/sample) instead of re-running the forward there
            out, zstash = _b.mlp_fwd_stash(flat_params, x, t, t_table)
            ctx.save_for_backward(x, t, flat_params, zstash)
            return out
        ctx.save_for_backward(x, t, flat_params)
        return _b.mlp_fwd(flat_params, x, t, precision, t_table)

    @staticmethod
    def backward(ctx, dout):
        x, t, flat_params, *rest = ctx.saved_tensors
        dparams = _b.mlp_bwd(flat_params, x, t, dout.contiguous(), ctx.precision, ctx.t_table,
                             zstash=rest[0] if rest else None)
        return None, None, dparams, None, None


class RotPredict(FlatParamsMixin, nn.Module):
    def __init__(self, d_model=65, out_type="rotmat", in_type="rotmat", precision="fp32"):
        super().__init__()
        self.in_type = in_type
        self.out_type = out_type
        if in_type != "rotmat" or d_model != 65:
            raise NotImplementedError("so3x: the fused score network is built for in_type='rotmat', d_model=65")
        if out_type not in ("skewvec", "rotmat"):
            raise ValueError(f"Unexpected out_type: {out_type}")  # the reference builds this error without raising it (so3_train.py:24)
        if precision not in _PRECISIONS:
            raise ValueError(f"precision must be one of {list(_PRECISIONS)}")
        self.precision = precision
        self.d_out = 3 if out_type == "skewvec" else 6  # "rotmat": six2rmat of a 6-wide head (so3_train.py:19-22, 47-48)
        self.time_embedding = SinusoidalPosEmb(d_model - 9)
        self.net = nn.Sequential(
            nn.Linear(d_model, d_model), nn.SiLU(),
            nn.Linear(d_model, d_model), nn.SiLU(),
            nn.Linear(d_model, d_model), nn.SiLU(),
            nn.Linear(d_model, d_model), nn.SiLU(),
            nn.Linear(d_model, self.d_out),
        )
        # 0: timesteps are arbitrary (embedding evaluated per sample in-kernel).  T > 0: the caller promises
        # 0 <= t < T; the kernels then gather per-timestep table rows instead of 56 sin/cos per sample.
        # SO3Diffusion passes its num_timesteps per call (forward's t_table argument); this attribute is the
        # default for direct calls and may be set by a caller that knows its timestep range.
        self.t_table = 0
        # "f16": the reverse chain (so3x_p_sample_chain) runs its bf16 path with IEEE half operand bits (precision "bf16" only; a
        # labelled extra leg of round 4 -- three more mantissa bits and one instruction less per activation; training is untouched)
        self.chain_operands = None
        # the 17,358 (skewvec) / 17,556 (rotmat) parameters live in ONE flat buffer in state_dict order; the nn.Linear
        # parameters are views of it (so3x.flat): flat_data() / flat_params() / flat_grad()
        self._init_flat()

    @property
    def precision_code(self) -> int:
        return _PRECISIONS[self.precision]

    @property
    def chain_precision_code(self) -> int:
        """the precision argument of the reverse-chain kernel: this network's, or the f16-operand leg of the bf16 path"""
        if self.chain_operands not in (None, "f16"):
            raise ValueError("chain_operands must be None or 'f16'")
        if self.chain_operands == "f16" and self.precision == "bf16":
            return _b.PREC_F16
        return _PRECISIONS[self.precision]

    def forward(self, x: torch.Tensor, t: torch.Tensor, t_table: int = None, raw: bool = False):
        """raw=True returns the network's [.., d_out] outputs without the six2rmat of out_type="rotmat" (SO3Diffusion's
        prevstep loss applies it inside its own fused kernel)"""
        tt = self.t_table if t_table is None else int(t_table)
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.net.parameters()):
            out = _ScoreMLPFn.apply(x, t, self.flat_params(), self.precision_code, tt)
        else:
            out = _b.mlp_fwd(self.flat_params_nograd(), x, t, self.precision_code, tt)
        return _b.six2rmat(out) if (self.out_type == "rotmat" and not raw) else out


def main(argv=None):
    """Training loop of the reference's so3_train.py:54-81 (two-mode toy data, Adam 3e-4), data-parallel over the GPUs of
    one node when launched with torchrun: the batch axis is sharded, the flat gradient is all-reduced once per step (RCCL)
    and, with --graph, the whole step -- noising, network, loss, backward, all-reduce, Adam -- replays as a captured
    hipGraph (so3x.graphs.TrainStepGraph)."""
    from .diffusion import SO3Diffusion
    from . import parallel
    from . import optim as so3x_optim

    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=BATCH, help="global batch (reference: 64)")
    ap.add_argument("--steps", type=int, default=400000)
    ap.add_argument("--timesteps", type=int, default=1000)
    ap.add_argument("--precision", default="fp32", choices=list(_PRECISIONS))
    ap.add_argument("--lr", type=float, default=3e-4)
    ap.add_argument("--log-every", type=int, default=10)
    ap.add_argument("--save-every", type=int, default=1000)
    ap.add_argument("--graph", action="store_true", help="replay the whole step as a captured hipGraph")
    ap.add_argument("--optimizer", default="so3x", choices=["so3x", "torch"],
                    help="so3x: Adam as one launch on the flat buffers (so3x.optim.Adam); torch: torch.optim.Adam(fused=True)")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--weights", default="weights/weights_so3.pt")
    args = ap.parse_args(argv)

    ctx = parallel.init()
    device = ctx.device
    torch.manual_seed(args.seed)  # same initial weights on every rank (and broadcast below)
    net = RotPredict(out_type="skewvec", precision=args.precision).to(device)
    net.train()
    parallel.broadcast_parameters(net, ctx)
    # the timesteps of p_losses come from torch's device generator (diffusion.py:373): decorrelate the ranks
    torch.cuda.manual_seed(args.seed + 7919 * (ctx.rank + 1))
    process = SO3Diffusion(net, timesteps=args.timesteps, loss_type="skewvec").to(device)
    if args.optimizer == "so3x":
        optim = so3x_optim.Adam(net, lr=args.lr)  # the reference's Adam (so3_train.py:64), one launch
    else:
        optim = torch.optim.Adam(net.parameters(), lr=args.lr, fused=True, capturable=args.graph)
    z90 = torch.tensor([[0.0, -1.0, 0.0], [1.0, 0.0, 0.0], [0.0, 0.0, 1.0]])
    rotations = torch.stack((z90, z90.T), dim=0).to(device)
    lo, hi = parallel.shard_range(args.batch, ctx.rank, ctx.world_size)
    process.index_base = lo
    gen = torch.Generator(device=device).manual_seed(1234 + ctx.rank)
    t0 = time.time()
    graph = None
    if args.graph:
        from .graphs import TrainStepGraph
        graph = TrainStepGraph(process, optim, (hi - lo, 3, 3), ctx=ctx, n_global=args.batch)
    for i in range(1, args.steps + 1):
        idx = torch.randint(0, 2, (hi - lo,), device=device, generator=gen)
        truepos = rotations[idx]
        if graph is not None:
            loss = graph.step(truepos)
        else:
            loss = process(truepos)
            optim.zero_grad()
            loss.backward()
            parallel.allreduce_gradients(net, ctx, n_local=hi - lo, n_global=args.batch, optimizer=optim)
            optim.step()
        if i % args.log_every == 0:
            lval = parallel.mean_scalar(loss.detach(), ctx)
            if ctx.rank == 0:
                print(json.dumps({"step": i, "loss": lval, "elapsed_s": round(time.time() - t0, 3)}), flush=True)
        if i % args.save_every == 0:
            if graph is not None:
                graph.flush()  # the pipelined step is one update behind: apply it before the parameters are read (all ranks: a collective)
            if ctx.rank == 0:
                os.makedirs(os.path.dirname(args.weights) or ".", exist_ok=True)
                torch.save(net.state_dict(), args.weights)
    if graph is not None:
        graph.flush()
    parallel.finalize(ctx)
    return net


if __name__ == "__main__":
    main()
