"""Flat parameter / gradient storage for the fused score networks.

The kernels of libso3x take the network's parameters as ONE contiguous fp32 buffer in state_dict order and return the
gradient the same way.  The two RotPredict modules therefore keep every nn.Linear parameter as a VIEW into one flat
tensor (state_dict keys, shapes and nn.Module behaviour are unchanged), and hand autograd's flat gradient back to the
parameters as views of one flat gradient tensor:

  * sampling always sees the current weights (no cached copy that an in-place optimizer update could leave stale);
  * a data-parallel step all-reduces `flat_grad()` in one call, with no cat / split / copy kernels;
  * `so3x.optim.Adam` updates `flat_data()` from `flat_grad()` in one launch."""
import torch

__all__ = ["FlatParamsMixin"]

# Every registration of a parameter or a submodule anywhere in the process bumps this counter (torch's global registration
# hooks: `module.weight = nn.Parameter(...)`, `net[0] = nn.Linear(...)`, load_state_dict(assign=True) all go through them).
# While it stands still, a network's cached parameter list IS its parameter list, and the per-step check (twice per training
# step) is ten pointer comparisons instead of a walk over the module tree (~20 us of a 280-us host-bound step).
_REGISTRATIONS = [0]


def _registered(*_args, **_kwargs):
    _REGISTRATIONS[0] += 1
    return None


_HOOKED = [False]


def _install_hooks():
    """the process-global registration hooks, installed at the first flattening (not at import: a process that only imports
    so3x pays nothing)"""
    if not _HOOKED[0]:
        torch.nn.modules.module.register_module_parameter_registration_hook(_registered)
        torch.nn.modules.module.register_module_module_registration_hook(_registered)
        _HOOKED[0] = True


# Parameter updates torch cannot see: a replayed hipGraph (so3x.graphs.TrainStepGraph) rewrites the flat buffer without bumping its
# tensor version.  Whoever does that calls params_changed_out_of_band(); caches derived from parameter values (the prepared
# sampling state of SO3Diffusion) key on (tensor version, this epoch).
PARAM_EPOCH = [0]


def params_changed_out_of_band():
    PARAM_EPOCH[0] += 1


class PreparedCache:
    """A kernel-side image derived from parameter VALUES (SO3Diffusion's prepared sampling state, PlaneNet's bf16 weight image),
    kept while the parameters do not change.  The cheap key -- (buffer address, every parameter's tensor version, PARAM_EPOCH,
    caller's extras) -- sees torch's in-place updates, load_state_dict and this package's graph replays.  It does NOT see writes
    through `p.data` (EMA / weight averaging), raw pointers or a user's own captured graph: those leave the versions alone.  For
    them the cache also keeps a device-side fingerprint of the flat buffer (int64 sum of its bit patterns, made at build time with
    no synchronisation) and compares it -- one small reduction + one host read -- whenever `suspicious` says so:
      * the caller asks (`check=True`: the start of a reverse chain, a module's train() / eval() switch, a new no_grad scope),
      * the cache sat idle for more than `idle_s` (a sampling loop calls back within microseconds; an update between two sampling
        runs does not),
      * every `every` hits regardless.
    A mismatch rebuilds.  `invalidate()` drops the image unconditionally.  What remains uncovered -- a `.data` write in the middle
    of a tight loop of hits -- is documented in INTEGRATION.md."""

    def __init__(self, every=256, idle_s=0.02):
        self.key, self.value, self.fp, self.hits, self.last, self.every, self.idle_s = None, None, None, 0, 0.0, every, idle_s
        self.force = True

    @staticmethod
    def fingerprint(flat):
        return flat.detach().view(torch.int32).sum(dtype=torch.int64)

    def invalidate(self):
        self.key = self.value = self.fp = None

    def check_next(self):
        """the next get() compares fingerprints (mode switches, chain starts)"""
        self.force = True

    def get(self, key, flat, build, check=False):
        import time
        now = time.monotonic()
        if self.key is not None and self.key == key:
            self.hits += 1
            if check or self.force or now - self.last > self.idle_s or self.hits % self.every == 0:
                if not bool((self.fingerprint(flat) == self.fp).item()):
                    self.key = None
            self.force = False
        if self.key is None or self.key != key:
            self.value, self.key, self.fp, self.hits, self.force = build(), key, self.fingerprint(flat), 0, False
        self.last = now
        return self.value


class _FlatView(torch.autograd.Function):
    """The flat buffer as a differentiable function of the individual parameters (what torch.cat(params) would be, with
    no copy either way).  backward: the flat gradient split into per-parameter views; in the usual loop (`zero_grad();
    loss.backward()`: every p.grad is None) the engine installs them as they are, so the .grads ARE one flat tensor
    (flat_grad() recognises it); accumulation, `backward(inputs=...)` and `torch.autograd.grad` get what they would from
    torch.cat."""

    @staticmethod
    def forward(ctx, net, *params):
        ctx.net = net
        return net._flat.detach()

    @staticmethod
    def backward(ctx, dflat):
        net = ctx.net
        if dflat.is_cuda and dflat.is_contiguous() and torch.cuda.is_current_stream_capturing() and all(
                p.grad is None and not p._backward_hooks and not p._post_accumulate_grad_hooks for p in net._flat_params if p.requires_grad):
            # (inside a graph capture the gradient is hung on the parameters here: see _FusedSkewvecLoss.backward)
            net._install_flat_grad(dflat)
            return (None,) * (1 + len(net._flat_params))
        return (None, *net._grad_views(dflat.contiguous()))


class FlatParamsMixin:
    """Mixed into an nn.Module whose parameters of `self.net` (or of `self._flat_root()`, if the class overrides it) live in one
    flat buffer."""

    def _flat_root(self):
        return self.net

    def _init_flat(self):
        self._flat = None          # the flat parameter buffer the nn.Parameters are views of
        self._flat_grad = None     # the flat gradient tensor the .grad attributes are views of (None until a backward)
        self._flat_params = []
        self._flatten()

    def _flatten(self):
        _install_hooks()
        params = list(self._flat_root().parameters())
        if not params:
            return
        dev, dt = params[0].device, params[0].dtype
        flat = torch.empty(sum(p.numel() for p in params), device=dev, dtype=dt)
        off = 0
        with torch.no_grad():
            for p in params:
                n = p.numel()
                flat[off:off + n].copy_(p.data.reshape(-1))
                p.data = flat[off:off + n].view(p.shape)
                off += n
        self._flat, self._flat_params, self._flat_grad, self._grad_cache = flat, params, None, None
        self._flat_owners = [(m, k, q) for m in self._flat_root().modules() for k, q in m._parameters.items() if q is not None]
        self._flat_seen = -1       # the registration count at which the cached list was last compared with the module tree

    def _flat_ok(self):
        """the cached list IS the module's current parameters (identity: load_state_dict(assign=True) or any replacement of an
        nn.Parameter re-homes them without touching the old objects) and each of them sits at its offset of the flat buffer"""
        f, ps = self._flat, self._flat_params
        if f is None or len(ps) == 0:
            return False
        if getattr(self, "_flat_seen", -1) == _REGISTRATIONS[0] and all(m._parameters.get(k) is q for m, k, q in self._flat_owners):
            # nothing has been registered anywhere since the list was last checked against the tree, and every owning module
            # still holds its parameter object under its name (torch's hooks do not fire for REMOVALS -- `del layer.weight`,
            # `register_parameter(name, None)`, edits of `_parameters` -- which this identity check catches; removing a whole
            # layer from the tree is not supported: the kernels are built for the network's fixed structure): only a caller
            # assigning p.data (which registers nothing) can have moved a parameter
            off, base, es = 0, f.data_ptr(), f.element_size()
            for p in ps:
                if p.data_ptr() != base + es * off:
                    return False
                off += p.numel()
            return True
        seen = _REGISTRATIONS[0]
        cur = list(self._flat_root().parameters())
        if len(cur) != len(ps) or any(a is not b for a, b in zip(cur, ps)):
            return False
        self._flat_seen = seen
        off, es = 0, f.element_size()
        for p in ps:
            if p.device != f.device or p.dtype != f.dtype or not p.is_contiguous() or p.data_ptr() != f.data_ptr() + es * off:
                return False
            off += p.numel()
        return off == f.numel()

    def _ensure_flat(self):
        # .to(device) / deepcopy / a caller assigning p.data re-home the parameters: adopt them again
        if not self._flat_ok():
            self._flatten()

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)
        self._flatten()
        return out

    def flat_data(self) -> torch.Tensor:
        """all parameters, flat, in state_dict order (no copy; always current)"""
        self._ensure_flat()
        return self._flat

    def flat_params_nograd(self) -> torch.Tensor:
        return self.flat_data()

    def flat_params(self) -> torch.Tensor:
        """the same buffer, differentiable: its gradient is routed to the nn.Linear parameters as views (no copy)"""
        self._ensure_flat()
        return _FlatView.apply(self, *self._flat_params)

    def _install_flat_grad(self, dflat):
        """the parameters' .grad become views of `dflat` (frozen parameters -- requires_grad False -- keep theirs untouched, as
        autograd would leave them)"""
        off = 0
        for p in self._flat_params:
            if p.requires_grad:
                p.grad = dflat[off:off + p.numel()].view(p.shape)
            off += p.numel()
        self._flat_grad = dflat

    def _grad_views(self, dflat):
        """per-parameter views of a flat gradient, for an autograd Function to return (None for frozen parameters); remembers
        the flat tensor, so that flat_grad() finds it again if the engine installs the views as the .grads"""
        off, out = 0, []
        for p in self._flat_params:
            out.append(dflat[off:off + p.numel()].view(p.shape) if p.requires_grad else None)
            off += p.numel()
        self._flat_grad = dflat
        return out

    def _persistent_grad(self):
        """(flat gradient buffer, its per-parameter views) for the training fast path's direct backward: allocated once per flat
        buffer and rewritten by every such backward, like the gradients of a loop that runs zero_grad(set_to_none=False) -- ten
        view constructions (~25 us of host time) less per step, and addresses a captured graph can rely on"""
        c = getattr(self, "_grad_cache", None)
        f = self._flat
        if c is None or c[0] is not f or c[1].device != f.device:
            g = torch.empty_like(f)
            views, off = [], 0
            for p in self._flat_params:
                views.append(g[off:off + p.numel()].view(p.shape))
                off += p.numel()
            c = self._grad_cache = (f, g, views)
        return c[1], c[2]

    def flat_grad(self):
        """the flat gradient if every parameter's .grad currently is a view of one flat tensor, else None"""
        g, ps = self._flat_grad, self._flat_params
        if g is None or not ps:
            return None
        off, es = 0, g.element_size()
        for p in ps:
            if p.requires_grad and (p.grad is None or p.grad.data_ptr() != g.data_ptr() + es * off):
                return None
            off += p.numel()
        return g

    def gather_flat_grad(self):
        """flat_grad(), or (gradients that arrived some other way) a flat copy installed as the .grad views"""
        g = self.flat_grad()
        if g is None:
            self._ensure_flat()
            g = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in self._flat_params])
            self._install_flat_grad(g)
        return g
