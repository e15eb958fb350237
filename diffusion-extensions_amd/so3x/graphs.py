"""hipGraph capture of a whole training step (forward noising + score network + loss + backward + optimizer).

At 2^19 samples the GPU side of a step is ~0.45 ms and the Python host path (autograd Functions, ctypes wrappers, the
optimizer) about the same, so the step is host-bound; replaying a captured graph takes the host out of the loop.  All
kernels of libso3x take an explicit stream, allocate nothing and never synchronise, so the step is capturable as it is;
the one host-side value that must change between replays -- the Philox offset of the noise draw -- moves into a device
counter (`SO3Diffusion.rng_counter`) that the graph itself increments, and `t` comes from torch's graph-safe generator."""
import torch

__all__ = ["TrainStepGraph"]


class TrainStepGraph:
    """process: SO3Diffusion (or a subclass whose p_losses honours `rng_counter`); optimizer: a capturable torch optimizer,
    e.g. torch.optim.Adam(params, lr, fused=True, capturable=True); batch_shape: the fixed shape of the data batch.

        g = TrainStepGraph(process, optim, x.shape)
        for x in data: loss = g.step(x)          # loss: 0-d device tensor, overwritten by the next replay

    If eager training steps ran before, drop every reference to their losses first (`del loss`): a live loss keeps the
    parameters' AccumulateGrad nodes bound to the stream of that earlier backward, and torch cannot capture across it.
    """

    def __init__(self, process, optimizer, batch_shape, warmup=3):
        self.process, self.optimizer = process, optimizer
        dev = process.betas.device
        self.x = torch.zeros(batch_shape, dtype=torch.float32, device=dev)
        self.x[..., 0, 0] = self.x[..., 1, 1] = self.x[..., 2, 2] = 1.0
        if process.rng_counter is None:
            process.rng_counter = torch.zeros(1, dtype=torch.int64, device=dev)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):  # warm-up off the default stream: one-time attribute / occupancy queries, allocator pools
            for _ in range(warmup):
                self._eager()
        torch.cuda.current_stream(dev).wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.loss = self._eager()

    def _eager(self):
        self.optimizer.zero_grad(set_to_none=True)
        loss = self.process(self.x)
        loss.backward()
        self.optimizer.step()
        return loss.detach()

    def step(self, x):
        self.x.copy_(x)
        self.graph.replay()
        return self.loss
