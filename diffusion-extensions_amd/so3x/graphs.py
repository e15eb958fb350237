"""hipGraph capture of a whole training step (forward noising + score network + loss + backward + gradient all-reduce +
optimizer; reference so3_train.py:69-81).

At 2^19 samples the GPU side of a step is ~0.24 ms and an eager Python step costs about as much host time, so the loop is
host-bound; replaying a captured graph takes the host out of it.  All kernels of libso3x take an explicit stream, allocate
nothing and never synchronise, so the step is capturable as it is; the one host-side value that must change between
replays -- the Philox offset of the noise draw -- lives in a device counter (`SO3Diffusion.rng_counter`) that the step
itself increments, `t` is drawn in the noising kernel, and so3x.optim.Adam keeps its step count on the device.

Three forms.

**One-kernel** (round 4; the default wherever the path below applies, any world size): so3x_train_fused -- noising, forward, loss
and the whole backward down to the dW slabs in ONE launch -- then the slab reduction with Adam in its epilogue (single process:
three launches per step with the prep), or slab reduction -> all-reduce -> Adam (data-parallel).  The noising is inside the
kernel, so there is nothing left to run beside the tail: the kernel is ~90 us shorter than the three it replaces, which is more
than the pipelined form below could hide.

**Pipelined** (round 3's default in a data-parallel run; since round 4 `pipeline=True` only -- "auto" takes the one-kernel step at
every world size, a choice that rests on ONE-GPU timings and gloo dry runs: no multi-GPU node was available to any round, so
which form wins over xGMI is unmeasured) for the path BASELINE config 4 names: SO3Diffusion + the 65-wide
skew-vector RotPredict with bf16 operands + so3x.optim.Adam).  The step runs as its C-ABI stages (so3x_train_noise / _net / _bwd_partial / _bwd_reduce /
so3x_adam_step, no autograd in between), and the graph boundary sits BEHIND THE FUSED BACKWARD instead of behind the optimizer:

    replay k:   side stream:  noise(batch k)                                      --+
                main stream:  reduce(k-1) -> all-reduce(k-1) -> Adam(k-1)         --+-> net(k) -> backward(k)

The noising of batch k depends on the data and the Philox counter only -- not on the parameters -- so it runs beside the tail
of step k-1: the slab reduction, the RCCL all-reduce of the 69 KB flat gradient and the optimizer, which are latency-bound and
leave the chip nearly empty ("overlapped with next step's noise generation").  Every kernel sees the inputs it
sees in the serial order, so losses and parameters are bit-identical to the eager loop.  The price is a one-step software
pipeline: after `step(x)` returns, the loss is batch k's, and the parameters carry the updates up to k-1; `flush()` runs the
outstanding tail (call it before reading / saving / sampling from the parameters; `step` after a flush starts the pipeline
again).

**Serial** (a single process, any other process / denoiser / optimizer, or `pipeline=False`): forward, backward (autograd),
all-reduce and optimizer captured as one stream, as in round 2.  With no collective the tail's kernels fill the chip for their
few microseconds and the noising finds no room beside them: measured, the pipelined form is 2-4 us SLOWER in a single process,
and `pipeline="staged"` -- the stages as one stream with the slab reduction and Adam as ONE launch
(`so3x_train_bwd_reduce_adam`), five launches, no autograd -- exactly as fast as this form.

Data-parallel (one process per GPU): `allreduce="in_graph"` captures the collective with the rest -- ONE graph launch per step;
`"split"` issues the collective eagerly between two graph launches, for stacks whose collectives cannot be captured (gloo);
`"auto"` tries the first and falls back to the second.  The choice is COLLECTIVE: every rank reports whether its capture worked
and all of them take the in-graph form only if all did (one MIN all-reduce), so no rank can replay a program with a different
sequence of collectives than its peers.

Construction leaves no trace: the warm-up steps that capture needs (allocator pools, communicator, attribute queries) run real
updates on a placeholder batch, so parameters, optimizer state, the Philox counters and torch's device generator are
snapshotted before and restored after -- the first replay is the first step of the eager loop, bit for bit."""
import copy

import torch
import torch.distributed as dist

from . import backend as _b
from . import parallel
from . import rng as _rng

__all__ = ["TrainStepGraph"]


def _hyper_of(optimizer):
    out = []
    for g in optimizer.param_groups:
        out.append(tuple((k, (float(v) if not isinstance(v, (tuple, list)) else tuple(float(a) for a in v)))
                         for k, v in sorted(g.items()) if k in ("lr", "betas", "eps", "weight_decay") and not isinstance(v, torch.Tensor)))
    return (tuple(out), float(getattr(optimizer, "grad_scale", 1.0)))


class TrainStepGraph:
    """process: SO3Diffusion (or a subclass whose p_losses honours `rng_counter`); optimizer: so3x.optim.Adam, or a
    capturable torch optimizer (torch.optim.Adam(params, lr, fused=True, capturable=True)); batch_shape: the fixed shape of
    this rank's data batch; ctx: so3x.parallel.Ctx (None = single process); n_global: the global batch (for unequal shards).
    pipeline: "auto" (the one-kernel step where the path allows, else the serial form), "fused" (the one-kernel step, or raise),
    True (round 3's pipelined stages), False (the serial form: autograd), "staged" (single process: round 3's stages in one stream,
    reduction + Adam as one launch).

        g = TrainStepGraph(process, optim, x.shape, ctx=ctx)
        for x in data: loss = g.step(x)          # loss: 0-d device tensor, overwritten by the next replay
        g.flush()                                # pipelined form: the last step's all-reduce + update

    Hyper-parameters (lr, betas, eps, weight_decay, grad_scale) are kernel arguments and therefore FROZEN at capture: a replay
    after they changed raises (re-create the graph; an LR schedule needs one graph per value).

    If eager training steps ran before, drop every reference to their losses first (`del loss`): a live loss keeps the
    parameters' AccumulateGrad nodes bound to the stream of that earlier backward, and torch cannot capture across it.
    """

    def __init__(self, process, optimizer, batch_shape, warmup=3, ctx=None, n_global=None, allreduce="auto", pipeline="auto",
                 check_every=100, _inject_capture_failure=False, _assume_capturable=False):
        self.process, self.optimizer = process, optimizer
        # every `check_every` calls of step() the loss is read back (one host synchronisation) and a non-finite value raises: the
        # one-kernel step reports an abandoned LDS hand-shake (a hung partner wave) as a NaN loss, and although its consumers then
        # leave parameters and optimizer state untouched (so3x_train_bwd_reduce{,_adam}, so3x_adam_step), a loop that never looks at
        # the loss would train on nothing without noticing.  0 = never check.
        self.check_every, self._steps = int(check_every), 0
        self.ctx = ctx
        self.world = 1 if ctx is None else ctx.world_size
        self.n_local = int(batch_shape[0])
        self.n_global = n_global
        self.net = process.denoise_fn
        dev = self.dev = process.betas.device
        if allreduce not in ("auto", "in_graph", "split"):
            raise ValueError("allreduce must be 'auto', 'in_graph' or 'split'")
        from .optim import Adam as _So3xAdam
        eligible = (process._lean(None) and process.draw_t_in_kernel and isinstance(optimizer, _So3xAdam) and optimizer.net is self.net
                    and self.n_local > 0 and all(p.requires_grad for p in self.net.net.parameters()))
        if pipeline is True and not eligible:
            raise ValueError("so3x: the pipelined step is built for SO3Diffusion(loss_type='skewvec', draw_t_in_kernel) + the 65-wide "
                             "skew-vector RotPredict with bf16 operands + so3x.optim.Adam, with EVERY parameter trainable (a frozen "
                             "parameter -- requires_grad False -- disqualifies it: the flat update would move it)")
        # "auto": pipelined where there is a collective to hide.  In a single process every kernel of the tail fills the chip
        # for its few microseconds, the noising kernel finds no free wave slots beside them, and the fork / join edges cost
        # ~2 us: measured 0.2430 vs 0.2402 ms per 2^19-sample step (bench.py train_step, round 3) -- so the serial form stays.
        if pipeline == "fused" and not eligible:
            raise ValueError("so3x: the one-kernel step is built for SO3Diffusion(loss_type='skewvec', draw_t_in_kernel) + the 65-wide "
                             "skew-vector RotPredict with bf16 operands + so3x.optim.Adam, with EVERY parameter trainable (a frozen "
                             "parameter -- requires_grad False -- disqualifies it: the flat update would move it)")
        # "auto": the one-kernel step (so3x_train_fused) wherever it applies -- measured 0.2066 against 0.2446 ms for the staged
        # launches at 2^19 samples in its first form (profiles/r04_ab_train_fused_v1.json)
        self.fused = bool(eligible and pipeline in ("auto", "fused") and getattr(process, "train_step_kernel", "fused") == "fused")
        self.pipelined = bool(eligible and pipeline is True)
        # ... decided TOGETHER (as in_graph / split below): the forms differ in their sequence of collectives (the pipelined one
        # flushes with an extra all-reduce), so a rank whose shard is empty must not take another form than its peers
        self.fused, self.pipelined = self._all_ranks(self.fused), self._all_ranks(self.pipelined)
        # pipeline="staged" (single process; A/B): the same stages as ONE stream with the slab reduction and the optimizer as one
        # launch (so3x_train_bwd_reduce_adam) -- five launches, no autograd.  Measured the same as the autograd-driven serial
        # graph (0.2376 against 0.2366 ms: inside a graph a 5-us launch costs nothing extra), so "auto" keeps the serial form.
        if pipeline == "staged" and not (eligible and self.world == 1):
            raise ValueError("so3x: pipeline='staged' is the single-process form of the path the pipelined step is built for")
        self.staged = pipeline == "staged"
        snap = self._snapshot()
        self.x = torch.zeros(batch_shape, dtype=torch.float32, device=dev)
        self.x[..., 0, 0] = self.x[..., 1, 1] = self.x[..., 2, 2] = 1.0
        if process.rng_counter is None:
            process.rng_counter = torch.zeros(1, dtype=torch.int64, device=dev)
        self._one = torch.ones((), dtype=torch.float32, device=dev)  # d loss / d loss: given, not filled by a launch per step
        self._pending = False      # pipelined: a backward's slabs are waiting for their reduction / all-reduce / update
        self._side = torch.cuda.Stream(device=dev)
        if self.pipelined or self.staged or self.fused:
            self.buf = _b.TrainBuffers(self.n_local, process.num_timesteps, dev, staged=not self.fused)
            self.net.flat_data()                         # (re-)adopt the parameters into the flat buffer if they were re-homed
            self.net._install_flat_grad(self.buf.grad)   # the .grad views alias the buffer the reduction writes
            self.loss = self.buf.loss[0]
            process._tables()
        try:   # whatever happens below, construction leaves no trace: warm-up updates, moments, counters are undone
            warm = torch.cuda.Stream(device=dev)
            warm.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(warm):   # warm-up off the default stream: one-time attribute / occupancy queries, allocator pools,
                for _ in range(warmup):     # optimizer state, and the communicator's first collective
                    if self.fused:
                        self._fused_step(); self._fused_tail()
                    elif self.pipelined:
                        self._noise(); self._net(); self._bwd(); self._tail()
                    elif self.staged:
                        self._noise(); self._net(); self._bwd(); self.optimizer.step_with_reduction(self.buf)
                    else:
                        self._fwd_bwd()
                        self._allreduce()
                        self.optimizer.step()
            torch.cuda.current_stream(dev).wait_stream(warm)
            self.mode = None
            self.graph = self.graph_opt = None
            if allreduce == "auto" and self.world > 1 and dist.get_backend() != "nccl" and not _assume_capturable:
                allreduce = "split"  # host-staged collectives (gloo) cannot be captured: do not try
            if self.world == 1 or allreduce in ("auto", "in_graph"):
                err = None
                try:
                    self._capture_in_graph(_inject_capture_failure)
                except Exception as e:  # noqa: BLE001 -- whatever the stack raises inside a capture
                    err = e
                    torch.cuda.synchronize(dev)
                # the decision is taken TOGETHER: one rank replaying [.. all-reduce ..] as a graph while another issues it eagerly
                # between two graphs would still match, but a rank that raised here while its peers went on would not
                if self._all_ranks(err is None):
                    self.mode = "in_graph"
                else:
                    self.graph = self.graph_head = self.graph_tail = None
                    if self.world == 1 or allreduce == "in_graph":
                        raise err if err is not None else RuntimeError("so3x: another rank could not capture the all-reduce inside the graph")
            if self.mode is None:
                self._capture_split()
                self.mode = "split"
        finally:
            self._restore(snap)
        self._hyper = _hyper_of(optimizer)

    # ------------------------------------------------------------------ state around the warm-up
    def _snapshot(self):
        dev = self.process.betas.device
        opt = self.optimizer
        snap = {"params": [p.detach().clone() for p in self.net.parameters()],
                "rng_counter": None if self.process.rng_counter is None else self.process.rng_counter.clone(),
                "host_rng": dict(_rng._state), "cuda_rng": torch.cuda.get_rng_state(dev),
                "grad_scale": getattr(opt, "grad_scale", None)}
        if hasattr(opt, "_m"):   # so3x.optim.Adam: flat moments + device step count (None until the first step)
            snap["so3x_adam"] = None if opt._m is None else (opt._m.clone(), opt._v.clone(), opt._step.clone())
        else:
            snap["opt_state"] = {p: {k: (v.clone() if isinstance(v, torch.Tensor) else copy.deepcopy(v)) for k, v in st.items()}
                                 for p, st in opt.state.items()}
        return snap

    @torch.no_grad()
    def _restore(self, snap):
        """in place: the captured graphs hold the addresses of these tensors"""
        torch.cuda.synchronize(self.dev)
        for p, q in zip(self.net.parameters(), snap["params"]):
            p.copy_(q)
        if snap["rng_counter"] is None:
            self.process.rng_counter.zero_()
        else:
            self.process.rng_counter.copy_(snap["rng_counter"])
        _rng._state.update(snap["host_rng"])
        torch.cuda.set_rng_state(snap["cuda_rng"], self.dev)
        opt = self.optimizer
        if "so3x_adam" in snap:
            if opt._m is not None:
                if snap["so3x_adam"] is None:
                    opt._m.zero_(); opt._v.zero_(); opt._step.zero_()
                else:
                    for dst, src in zip((opt._m, opt._v, opt._step), snap["so3x_adam"]):
                        dst.copy_(src)
        else:
            for p, st in opt.state.items():
                old = snap["opt_state"].get(p)
                for k, v in st.items():
                    if isinstance(v, torch.Tensor):
                        v.zero_() if old is None else v.copy_(old[k])
        if self.pipelined or self.staged or self.fused:
            self.buf.grad.zero_(); self.buf.loss.zero_()
        self._pending = False
        torch.cuda.synchronize(self.dev)

    def _all_ranks(self, ok: bool) -> bool:
        """True on every rank iff `ok` on every rank (one MIN all-reduce, outside any capture)"""
        if self.world == 1:
            return ok
        f = torch.tensor([1.0 if ok else 0.0], dtype=torch.float32, device=self.dev)
        dist.all_reduce(f, op=dist.ReduceOp.MIN)
        return bool(f.item() > 0)

    # ------------------------------------------------------------------ the step's pieces
    def _fwd_bwd(self):
        self.optimizer.zero_grad(set_to_none=True)
        loss = self.process(self.x)
        loss.backward(self._one)
        return loss.detach()

    def _allreduce(self):
        if self.world > 1:
            if self.pipelined or self.fused:
                parallel.allreduce_flat(self.buf.grad, self.ctx, n_local=self.n_local, n_global=self.n_global, optimizer=self.optimizer)
            else:
                parallel.allreduce_gradients(self.net, self.ctx, n_local=self.n_local, n_global=self.n_global, optimizer=self.optimizer)

    # the one-kernel step (so3x.h: so3x_train_fused) and what follows it
    def _fused_step(self):
        p = self.process
        _b.train_fused(self.buf, self.net.flat_data(), p._sched, p._trap_q, self.x, None, quirk_col0=p.quirk_col0, seed=_rng.seed(),
                       rng_offset=0, rng_counter=p.rng_counter, index_base=p.index_base, guide_q=p._guide_q)

    def _fused_tail(self):
        if self.world == 1:
            self.optimizer.step_with_reduction(self.buf)   # slab reduction + Adam: one launch
        else:
            self._tail()                                   # slab reduction -> all-reduce -> Adam

    # the stages of the pipelined form (so3x.h: so3x_train_noise / _net / _bwd_partial / _bwd_reduce, so3x_adam_step)
    def _noise(self):
        p = self.process
        _b.train_noise(self.buf, p._sched, p._trap_q, self.x, None, quirk_col0=p.quirk_col0, seed=_rng.seed(), rng_offset=0,
                       rng_counter=p.rng_counter, index_base=p.index_base, guide_q=p._guide_q)

    def _net(self):
        _b.train_net(self.buf, self.net.flat_data(), rng_counter=self.process.rng_counter)

    def _bwd(self):
        _b.train_bwd_partial(self.buf)

    def _tail(self):
        _b.train_bwd_reduce(self.buf)
        self._allreduce()
        self.optimizer.step()

    def _fork_noise(self):
        """noise(batch k) on the side stream, forked from and (by the caller) joined back into the capturing stream"""
        cur = torch.cuda.current_stream(self.dev)
        self._side.wait_stream(cur)
        with torch.cuda.stream(self._side):
            self._noise()
        return cur

    # ------------------------------------------------------------------ capture
    def _capture_in_graph(self, inject_failure=False):
        if self.fused:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self._fused_step()
                if inject_failure:
                    raise RuntimeError("so3x: injected capture failure (test)")
                self._fused_tail()
            self.graph = g
            return
        if self.staged:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self._noise(); self._net(); self._bwd()
                if inject_failure:
                    raise RuntimeError("so3x: injected capture failure (test)")
                self.optimizer.step_with_reduction(self.buf)
            self.graph = g
            return
        if not self.pipelined:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self.loss = self._fwd_bwd()
                self._allreduce()
                if inject_failure:
                    raise RuntimeError("so3x: injected capture failure (test)")
                self.optimizer.step()
            self.graph = g
            return
        head, steady, tail = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        with torch.cuda.graph(head):                       # first step of a pipeline: nothing is pending
            self._noise(); self._net(); self._bwd()
        with torch.cuda.graph(steady, pool=head.pool()):   # tail of step k-1 beside the noising of batch k, then net + backward of k
            cur = self._fork_noise()
            self._tail()
            cur.wait_stream(self._side)
            if inject_failure:
                raise RuntimeError("so3x: injected capture failure (test)")
            self._net(); self._bwd()
        with torch.cuda.graph(tail, pool=head.pool()):     # flush: the last step's reduction, all-reduce and update
            self._tail()
        self.graph_head, self.graph, self.graph_tail = head, steady, tail

    def _capture_split(self):
        if self.fused:   # [one-kernel step, slab reduction] -> eager all-reduce -> [Adam]
            g, g2 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self._fused_step()
                _b.train_bwd_reduce(self.buf)
            with torch.cuda.graph(g2, pool=g.pool()):
                self.optimizer.step()
            self.graph, self.graph_opt = g, g2
            return
        if not self.pipelined:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self.loss = self._fwd_bwd()
            g2 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g2, pool=g.pool()):  # the gradient buffer of the first graph is read by the second
                self.optimizer.step()
            self.graph, self.graph_opt = g, g2
            return
        head, red, rest, opt = (torch.cuda.CUDAGraph() for _ in range(4))
        with torch.cuda.graph(head):
            self._noise(); self._net(); self._bwd()
        with torch.cuda.graph(red, pool=head.pool()):      # [reduce]  -> eager all-reduce ->  [noise || Adam, net, backward]
            _b.train_bwd_reduce(self.buf)
        with torch.cuda.graph(rest, pool=head.pool()):
            cur = self._fork_noise()
            self.optimizer.step()
            cur.wait_stream(self._side)
            self._net(); self._bwd()
        with torch.cuda.graph(opt, pool=head.pool()):
            self.optimizer.step()
        self.graph_head, self.graph_red, self.graph, self.graph_opt = head, red, rest, opt

    # ------------------------------------------------------------------ replay
    def _check(self):
        self._steps += 1
        if self.check_every > 0 and self._steps % self.check_every == 0 and not bool(torch.isfinite(self.loss).all().item()):
            raise _b.So3xError(f"so3x: the training step's loss is not finite after {self._steps} steps (loss = {float(self.loss)}): a "
                               "hand-shake inside the one-kernel step gave up, or the data / parameters hold NaN; the update of such a "
                               "step is skipped on the device")

    def step(self, x):
        self.x.copy_(x)
        self.replay()
        self._check()
        return self.loss

    def replay(self):
        from .flat import params_changed_out_of_band
        params_changed_out_of_band()   # the captured update rewrites the parameters without touching their tensor version
        if _hyper_of(self.optimizer) != self._hyper:
            raise RuntimeError("so3x: optimizer hyper-parameters changed after capture; they are kernel arguments frozen into the graph "
                               "-- build a new TrainStepGraph")
        if not self.pipelined:
            self.graph.replay()
            if self.graph_opt is not None:
                self._allreduce()
                self.graph_opt.replay()
            return
        if not self._pending:
            self.graph_head.replay()
        elif self.mode == "in_graph":
            self.graph.replay()
        else:
            self.graph_red.replay()
            self._allreduce()
            self.graph.replay()
        self._pending = True

    def flush(self):
        """pipelined form: run the outstanding reduction + all-reduce + optimizer update (a no-op otherwise).  After it the
        parameters are those of the eager loop after the same number of steps."""
        from .flat import params_changed_out_of_band
        params_changed_out_of_band()
        if not (self.pipelined and self._pending):
            return
        if self.mode == "in_graph":
            self.graph_tail.replay()
        else:
            self.graph_red.replay()
            self._allreduce()
            self.graph_opt.replay()
        self._pending = False
