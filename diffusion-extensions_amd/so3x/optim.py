"""The optimizer of the reference's training loops -- torch.optim.Adam(net.parameters(), lr=3e-4) (reference
so3_train.py:64, so3_lock_train.py:70) -- as ONE launch on the score network's flat parameter / gradient buffers
(so3x_adam_step), with the step count on the device so that a captured hipGraph of the training step advances it itself.

Same hyper-parameters, defaults and update rule as torch.optim.Adam (amsgrad and maximize are not provided); pinned
against torch's own results in tests/golden/adam.npz."""
import torch
import torch.optim.optimizer as _torch_optimizer

from . import backend as _b
from .flat import FlatParamsMixin

__all__ = ["Adam"]


def _global_step_hooks():
    return getattr(_torch_optimizer, "_global_optimizer_pre_hooks", None) or getattr(_torch_optimizer, "_global_optimizer_post_hooks", None)


class Adam(torch.optim.Optimizer):
    """Adam over a so3x score network (so3_train.RotPredict / so3_lock_train.RotPredict).

        optim = so3x.optim.Adam(net, lr=3e-4)
        loss = process(x); optim.zero_grad(); loss.backward(); optim.step()

    grad_scale multiplies the gradient inside the update (1/world_size after a summed all-reduce costs no launch)."""

    def __init__(self, net, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        if not isinstance(net, FlatParamsMixin):
            raise TypeError("so3x.optim.Adam takes a so3x score network (flat parameter storage); use torch.optim.Adam otherwise")
        if lr < 0 or eps < 0 or not (0 <= betas[0] < 1) or not (0 <= betas[1] < 1) or weight_decay < 0:
            raise ValueError("invalid Adam hyper-parameter")
        super().__init__(list(net.net.parameters()), dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.net = net
        self.grad_scale = 1.0
        self._m = self._v = self._step = None
        self._calls = 0          # update launches issued through this object (step / step_with_reduction); see skipped_steps

    def _state(self, flat):
        if self._m is None or self._m.device != flat.device or self._m.numel() != flat.numel():
            self._m = torch.zeros_like(flat)
            self._v = torch.zeros_like(flat)
            self._step = torch.zeros(2, dtype=torch.float32, device=flat.device)  # [count, scratch]
        return self._m, self._v, self._step

    def _frozen_slices(self):
        """[offset, end) of every parameter with requires_grad False.  torch.optim.Adam leaves such parameters (grad None)
        untouched; the one-launch update runs over the whole flat buffer, so their values and moments are put back after it."""
        out, off = [], 0
        for p in self.net._flat_params:
            if not p.requires_grad:
                out.append((off, off + p.numel()))
            off += p.numel()
        return out

    @property
    def step_count(self) -> int:
        return 0 if self._step is None else int(self._step[0].item())

    def skipped_steps(self) -> int:
        """Updates this optimizer LAUNCHED but the device did not apply: the update kernels leave parameters, moments and the step
        count untouched when the gradient's first entry is not finite -- the all-NaN gradient of a training step whose in-kernel
        hand-shake gave up (so3x.h) -- where torch.optim.Adam would write the NaN into the parameters.  A run that has genuinely
        diverged therefore shows up HERE (and in its loss), not as NaN parameters: check it at your log cadence (one host read).
        Launches replayed from a captured graph are not counted (TrainStepGraph checks the loss itself, `check_every`)."""
        return self._calls - self.step_count()

    def step(self, closure=None):
        """torch.optim.Adam.step().  Not wrapped by torch.optim.Optimizer's profiling hook (see `hooked` below): step pre/post
        hooks registered on this optimizer (or globally) are honoured by taking torch's wrapper whenever any exists -- the common
        case of none costs nothing (the wrapper and its record_function are ~25 us of a host-bound step)."""
        if self._optimizer_step_pre_hooks or self._optimizer_step_post_hooks or _global_step_hooks():
            return self._hooked_step(closure)
        return self._step_impl(closure)

    step.hooked = True   # torch.optim.Optimizer._patch_step_function: leave this class's step as it is

    def _hooked_step(self, closure=None):
        fn = type(self).__dict__.get("_wrapped_step")
        if fn is None:
            fn = type(self)._wrapped_step = torch.optim.Optimizer.profile_hook_step(type(self)._step_impl)
        return fn(self, closure)

    def _step_impl(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        with torch.no_grad():
            flat = self.net.flat_data()
            grad = self.net.gather_flat_grad()
            m, v, step = self._state(flat)
            g = self.param_groups[0]
            frozen = [(a, b, flat[a:b].clone(), m[a:b].clone(), v[a:b].clone()) for a, b in self._frozen_slices()]
            _b.adam_step(flat, grad, m, v, step, g["lr"], g["betas"][0], g["betas"][1], g["eps"], g["weight_decay"], self.grad_scale)
            if not torch.cuda.is_current_stream_capturing():
                self._calls += 1
            for a, b, pf, pm, pv in frozen:
                flat[a:b].copy_(pf); m[a:b].copy_(pm); v[a:b].copy_(pv)
        return loss

    def zero_grad(self, set_to_none: bool = True):
        """torch.optim.Optimizer.zero_grad for the flat layout: with set_to_none (torch's default) ten attribute stores"""
        if not set_to_none:
            return super().zero_grad(set_to_none=False)
        for p in self.net._flat_params:
            p.grad = None
        self.net._flat_grad = None
        for group in self.param_groups:       # parameters handed in some other way (none in the reference's loops)
            for p in group["params"]:
                if p.grad is not None:
                    p.grad = None

    @torch.no_grad()
    def step_with_reduction(self, buf):
        """the slab reduction of a staged training step (so3x.backend.TrainBuffers) and this optimizer's update as ONE launch --
        what a single-process captured step uses instead of train_bwd_reduce + step() (same arithmetic, bit-identical)"""
        flat = self.net.flat_data()
        if self._frozen_slices():
            raise ValueError("so3x.optim.Adam.step_with_reduction updates every parameter; with frozen parameters use step()")
        m, v, step = self._state(flat)
        g = self.param_groups[0]
        _b.train_bwd_reduce_adam(buf, flat, m, v, step, g["lr"], g["betas"][0], g["betas"][1], g["eps"], g["weight_decay"], self.grad_scale)
        if not torch.cuda.is_current_stream_capturing():
            self._calls += 1

    def state_dict(self):
        return {"exp_avg": self._m, "exp_avg_sq": self._v, "step": self._step, "param_groups": [dict((k, v) for k, v in g.items() if k != "params")
                                                                                                for g in self.param_groups]}

    def load_state_dict(self, sd):
        flat = self.net.flat_data()
        m, v, step = self._state(flat)
        if sd.get("exp_avg") is not None:
            m.copy_(sd["exp_avg"]); v.copy_(sd["exp_avg_sq"]); step.copy_(sd["step"])
            self._calls = self.step_count()
        for g, s in zip(self.param_groups, sd.get("param_groups", [])):
            g.update(s)
