"""Data-parallel plumbing: one process per GPU, batch axis sharded, one RCCL all-reduce
of the flat 17,358-float gradient per training step.  Sampling
needs no collective: samples are independent and the Philox streams are keyed by the
GLOBAL sample index, so results are identical for any world size."""
import os
from dataclasses import dataclass

import torch
import torch.distributed as dist

__all__ = ["Ctx", "init", "finalize", "shard_range", "broadcast_parameters", "allreduce_gradients", "allreduce_flat", "mean_scalar",
           "any_rank_true", "sharded_p_sample_loop"]


@dataclass
class Ctx:
    rank: int
    world_size: int
    local_rank: int
    device: torch.device
    owns_pg: bool = False


def shard_range(n: int, rank: int, world_size: int):
    """Contiguous [lo, hi) of rank's share of n samples; remainders go to the low ranks."""
    base, rem = divmod(int(n), int(world_size))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def init(backend: str = None, device: str = None) -> Ctx:
    """Reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (torchrun).  backend defaults to
    'nccl' (= RCCL on ROCm) on GPUs; 'gloo' is used by the CPU tests of this plumbing."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if device is None:
        if torch.cuda.is_available():
            ndev = torch.cuda.device_count()
            if local >= ndev and world > 1 and (backend or os.environ.get("SO3X_DIST_BACKEND") or "nccl") == "nccl":
                raise RuntimeError(f"so3x: LOCAL_RANK {local} but {ndev} GPU(s) visible: RCCL needs one GPU per rank "
                                   "(ranks may share a device only over SO3X_DIST_BACKEND=gloo, for testing the plumbing)")
            device = f"cuda:{local % ndev}"
        else:
            device = "cpu"
    dev = torch.device(device)
    if dev.type == "cuda":
        torch.cuda.set_device(dev)
    owns = False
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        # SO3X_DIST_BACKEND: override for exercising the multi-rank plumbing where RCCL cannot run (e.g. two ranks sharing the
        # one GPU of a test box, LOCAL_RANK=0 for both, over gloo)
        backend = backend or os.environ.get("SO3X_DIST_BACKEND") or ("nccl" if dev.type == "cuda" else "gloo")
        dist.init_process_group(backend, rank=rank, world_size=world)
        owns = True
    return Ctx(rank, world, local, dev, owns)


def finalize(ctx: Ctx):
    if ctx.owns_pg and dist.is_initialized():
        dist.destroy_process_group()


def broadcast_parameters(module: torch.nn.Module, ctx: Ctx, src: int = 0):
    """rank `src`'s parameters to every rank.  The so3x score networks keep theirs in one flat buffer (so3x.flat): one
    broadcast straight into it, nothing to copy back and no cached flat copy that could go stale."""
    if ctx.world_size == 1:
        return
    flat_data = getattr(module, "flat_data", None)
    if flat_data is not None:
        dist.broadcast(flat_data(), src)
        return
    flat = torch.cat([p.data.reshape(-1) for p in module.parameters()])
    dist.broadcast(flat, src)
    off = 0
    with torch.no_grad():
        for p in module.parameters():
            n = p.numel()
            p.copy_(flat[off:off + n].view_as(p))  # in place, through the tensor: bumps its version counter
            off += n


def _is_nccl():
    return dist.get_backend() == "nccl"


def allreduce_flat(flat: torch.Tensor, ctx: Ctx, n_local: int = None, n_global: int = None, optimizer=None):
    """In-place mean of a flat gradient over the ranks: ONE collective (RCCL over xGMI on GPUs; 69 KB for the 65-wide network).

    Equal shards: the mean of the per-rank mean-losses is the global mean loss (diffusion.py:357 averages over B*3 elements).
    With RCCL the division rides in the collective (ReduceOp.AVG); otherwise it is folded into the optimizer update when
    `optimizer` has a `grad_scale` (so3x.optim.Adam), else one mul_.  Unequal shards (n_global % world != 0): every rank's
    gradient is weighted by its share n_local / n_global before a SUM."""
    if ctx.world_size == 1:
        return flat
    if n_local is not None and n_global is not None and n_local * ctx.world_size != n_global:
        flat.mul_(float(n_local) / float(n_global))
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        return flat
    if _is_nccl():
        dist.all_reduce(flat, op=dist.ReduceOp.AVG)
    else:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        if optimizer is not None and hasattr(optimizer, "grad_scale"):
            optimizer.grad_scale = 1.0 / ctx.world_size
        else:
            flat.mul_(1.0 / ctx.world_size)
    return flat


def allreduce_gradients(module: torch.nn.Module, ctx: Ctx, n_local: int = None, n_global: int = None, optimizer=None):
    """Average the gradients of `module` over the ranks with one flat all-reduce.  The so3x score networks hold their
    gradient as one flat tensor already (the fused backward returns it that way and the .grad attributes are views of it):
    no cat, no split, no copies."""
    if ctx.world_size == 1:
        return
    gather = getattr(module, "gather_flat_grad", None)
    if gather is not None:
        allreduce_flat(gather(), ctx, n_local, n_global, optimizer)
        return
    params = [p for p in module.parameters() if p.grad is not None]
    flat = torch.cat([p.grad.reshape(-1) for p in params])
    allreduce_flat(flat, ctx, n_local, n_global)
    off = 0
    for p in params:
        n = p.numel()
        p.grad.copy_(flat[off:off + n].view_as(p.grad))
        off += n


def any_rank_true(flag: torch.Tensor, ctx: Ctx) -> bool:
    """True on every rank if `flag` (a 0-d bool / number tensor) is set on any rank: one MAX all-reduce, so that all ranks
    take the same branch and keep entering the same collectives."""
    f = flag.detach().to(torch.float32).reshape(1).clone()
    if ctx.world_size > 1:
        dist.all_reduce(f, op=dist.ReduceOp.MAX)
    return bool(f.item() > 0)


def mean_scalar(x: torch.Tensor, ctx: Ctx) -> float:
    if ctx.world_size > 1:
        x = x.clone()
        dist.all_reduce(x, op=dist.ReduceOp.SUM)
        x = x / ctx.world_size
    return float(x.item())


@torch.no_grad()
def sharded_p_sample_loop(process, n_global: int, ctx: Ctx, gather: bool = False, x_init: torch.Tensor = None):
    """The full reverse chain for `n_global` rotations, batch-sharded over the ranks: rank r runs samples
    [lo, hi) with the Philox counters keyed by the GLOBAL sample index, so the assembled result is the same tensor for any
    number of ranks; no collective inside the chain.  Returns this rank's shard, or (gather=True) the whole [n_global, 3, 3]
    tensor on every rank via one all_gather.  x_init: optional [n_global, 3, 3] start (each rank takes its rows)."""
    lo, hi = shard_range(n_global, ctx.rank, ctx.world_size)
    saved = process.index_base
    process.index_base = lo
    try:
        x = process.p_sample_loop((hi - lo,), x_init=None if x_init is None else x_init[lo:hi].contiguous())
    finally:
        process.index_base = saved
    if not gather or ctx.world_size == 1:
        return x
    sizes = [shard_range(n_global, r, ctx.world_size) for r in range(ctx.world_size)]
    width = max(b - a for a, b in sizes)
    pad = torch.zeros((width, 3, 3), dtype=x.dtype, device=x.device)
    pad[: hi - lo] = x
    parts = [torch.empty_like(pad) for _ in range(ctx.world_size)]
    dist.all_gather(parts, pad)
    return torch.cat([parts[r][: b - a] for r, (a, b) in enumerate(sizes)], dim=0)
