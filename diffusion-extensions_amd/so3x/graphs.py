"""hipGraph capture of a whole training step (forward noising + score network + loss + backward + gradient all-reduce +
optimizer).

At 2^19 samples the GPU side of a step is ~0.3 ms and an eager Python step costs about as much host time, so the loop is
host-bound; replaying a captured graph takes the host out of it.  All kernels of libso3x take an explicit stream, allocate
nothing and never synchronise, so the step is capturable as it is; the one host-side value that must change between
replays -- the Philox offset of the noise draw -- lives in a device counter (`SO3Diffusion.rng_counter`) that the step
itself increments, `t` comes from torch's graph-safe generator, and so3x.optim.Adam keeps its step count on the device.

Data-parallel (one process per GPU): the flat-gradient all-reduce (RCCL) sits between the backward and the optimizer.
`allreduce="in_graph"` captures it with the rest -- ONE graph launch per step; `"split"` replays [forward + backward],
issues the collective eagerly on the same stream, and replays [optimizer] -- two graph launches and one RCCL call per step,
for stacks whose collectives cannot be captured.  `"auto"` (default) tries the first and falls back to the second."""
import torch
import torch.distributed as dist

from . import parallel

__all__ = ["TrainStepGraph"]


class TrainStepGraph:
    """process: SO3Diffusion (or a subclass whose p_losses honours `rng_counter`); optimizer: so3x.optim.Adam, or a
    capturable torch optimizer (torch.optim.Adam(params, lr, fused=True, capturable=True)); batch_shape: the fixed shape of
    this rank's data batch; ctx: so3x.parallel.Ctx (None = single process); n_global: the global batch (for unequal shards).

        g = TrainStepGraph(process, optim, x.shape, ctx=ctx)
        for x in data: loss = g.step(x)          # loss: 0-d device tensor, overwritten by the next replay

    If eager training steps ran before, drop every reference to their losses first (`del loss`): a live loss keeps the
    parameters' AccumulateGrad nodes bound to the stream of that earlier backward, and torch cannot capture across it.
    """

    def __init__(self, process, optimizer, batch_shape, warmup=3, ctx=None, n_global=None, allreduce="auto"):
        self.process, self.optimizer = process, optimizer
        self.ctx = ctx
        self.world = 1 if ctx is None else ctx.world_size
        self.n_local = int(batch_shape[0])
        self.n_global = n_global
        self.net = process.denoise_fn
        dev = process.betas.device
        self.x = torch.zeros(batch_shape, dtype=torch.float32, device=dev)
        self.x[..., 0, 0] = self.x[..., 1, 1] = self.x[..., 2, 2] = 1.0
        if process.rng_counter is None:
            process.rng_counter = torch.zeros(1, dtype=torch.int64, device=dev)
        self._one = torch.ones((), dtype=torch.float32, device=dev)  # d loss / d loss: given, not filled by a launch per step
        if allreduce not in ("auto", "in_graph", "split"):
            raise ValueError("allreduce must be 'auto', 'in_graph' or 'split'")
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):  # warm-up off the default stream: one-time attribute / occupancy queries, allocator pools,
            for _ in range(warmup):    # optimizer state, and the communicator's first collective
                self._fwd_bwd()
                self._allreduce()
                self.optimizer.step()
        torch.cuda.current_stream(dev).wait_stream(side)
        self.mode = None
        self.graph = self.graph_opt = None
        if allreduce == "auto" and self.world > 1 and dist.get_backend() != "nccl":
            allreduce = "split"  # host-staged collectives (gloo) cannot be captured: do not try
        if self.world == 1 or allreduce in ("auto", "in_graph"):
            try:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    self.loss = self._fwd_bwd()
                    self._allreduce()
                    self.optimizer.step()
                self.graph, self.mode = g, "in_graph"
            except Exception:
                if self.world == 1 or allreduce == "in_graph":
                    raise
                torch.cuda.synchronize(dev)
        if self.mode is None:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self.loss = self._fwd_bwd()
            g2 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g2, pool=g.pool()):  # the gradient buffer of the first graph is read by the second
                self.optimizer.step()
            self.graph, self.graph_opt, self.mode = g, g2, "split"

    def _fwd_bwd(self):
        self.optimizer.zero_grad(set_to_none=True)
        loss = self.process(self.x)
        loss.backward(self._one)
        return loss.detach()

    def _allreduce(self):
        if self.world > 1:
            parallel.allreduce_gradients(self.net, self.ctx, n_local=self.n_local, n_global=self.n_global, optimizer=self.optimizer)

    def step(self, x):
        self.x.copy_(x)
        self.replay()
        return self.loss

    def replay(self):
        self.graph.replay()
        if self.graph_opt is not None:
            self._allreduce()
            self.graph_opt.replay()
