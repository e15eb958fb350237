"""SO3Diffusion with the reference's constructor, buffers and methods
(reference diffusion.py:280-374 and the schedule buffers of GaussianDiffusion.__init__,
diffusion.py:57-92), running on the fused HIP kernels of libso3x.

Fast paths
  * p_losses          -> one launch: noise draw + q_sample + regression target
  * p_sample          -> one launch: score MLP + posterior mean + noise (when denoise_fn
                         is a so3x RotPredict); generic denoise_fn: MLP in torch, rest fused
  * p_sample_loop     -> ONE launch for the whole T-step chain (rotations stay in VGPRs)
Reference behaviours kept by default and switchable: the column-0 CDF gather of the
batched-eps sampler (`quirk_col0`), IGSO3(eps=1) chain initialisation.
"""
import weakref

import numpy as np
import torch
import torch.nn as nn

from . import backend as _b
from . import rng as _rng
from .distributions import IsotropicGaussianSO3
from .so3_train import RotPredict

__all__ = ["SO3Diffusion", "ProjectedSO3Diffusion", "cosine_beta_schedule", "extract", "noise_like"]

_SCHED_NAMES = ("betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_alphas_cumprod",
                "sqrt_one_minus_alphas_cumprod", "log_one_minus_alphas_cumprod", "sqrt_recip_alphas_cumprod",
                "sqrt_recipm1_alphas_cumprod", "posterior_variance", "posterior_log_variance_clipped",
                "posterior_mean_coef1", "posterior_mean_coef2")


def cosine_beta_schedule(timesteps, s=0.008):
    """float64 betas of the un-vendored denoising_diffusion_pytorch helper the reference imports
    (reference diffusion.py:8-14, 60).  Parity with the reference's fork is unpinned; pass
    `betas=` explicitly to remove the doubt.  The reference only ever calls it with the default offset s (diffusion.py:60):
    that case runs in libso3x (so3x_cosine_beta_schedule); any other s evaluates the same published formula in numpy
    float64 -- init-time host arithmetic, as in the reference."""
    if s == 0.008:
        return _b.cosine_beta_schedule(int(timesteps))
    steps = int(timesteps) + 1
    x = np.linspace(0, steps, steps)
    ac = np.cos(((x / steps) + s) / (1 + s) * np.pi * 0.5) ** 2
    ac = ac / ac[0]
    return np.clip(1 - (ac[1:] / ac[:-1]), a_min=0, a_max=0.999)


def extract(a, t, x_shape):
    """a[t] reshaped to broadcast against x_shape (the lucidrains helper)."""
    b = t.shape[0]
    return a.gather(-1, t).reshape(b, *((1,) * (len(x_shape) - 1)))


def noise_like(shape, device, repeat=False):
    """Gaussian noise of a shape, optionally one draw repeated along the batch (reference diffusion.py:19-22)"""
    if repeat:
        return torch.randn((1, *shape[1:]), device=device).repeat(shape[0], *((1,) * (len(shape) - 1)))
    return torch.randn(shape, device=device)


class _FusedStep:
    """One evaluation of the one-kernel training step: the buffers its backward half (the slab reduction) still has to read."""
    __slots__ = ("buf", "net", "staged", "carry", "T", "n_params")

    def direct_ok(self):
        """`loss.backward()` may skip the autograd engine: nothing is accumulated, frozen or hooked, so the engine would do exactly
        one thing -- call the slab reduction and hang its result on the parameters"""
        if self.buf is None or torch.is_anomaly_enabled():
            return False
        for p in self.net._flat_params:
            if p.grad is not None or not p.requires_grad or p._backward_hooks or p._post_accumulate_grad_hooks:
                return False
        return True

    def accumulate_ok(self):
        """a plain backward() whose only complications are gradients to add to and frozen parameters"""
        if self.buf is None or torch.is_anomaly_enabled():
            return False
        return not any(p._backward_hooks or p._post_accumulate_grad_hooks for p in self.net._flat_params if p.requires_grad)

    def run_accumulate(self):
        """what AccumulateGrad does, parameter by parameter: install the view where there is no gradient yet, add in place where
        there is one, leave frozen parameters alone"""
        net, buf = self.net, self.buf
        self.buf = None
        grad = _b.train_bwd_reduce(buf, gscale=None)
        _b.TrainBuffers.release(buf)
        with torch.no_grad():
            for p, v in zip(net._flat_params, net._grad_views(grad)):
                if v is None:
                    continue
                if p.grad is None:
                    p.grad = v
                else:
                    p.grad.add_(v)

    def run_direct(self):
        net, buf = self.net, self.buf
        self.buf = None
        g, views = net._persistent_grad()
        _b.train_bwd_reduce(buf, gscale=None, grad=g)
        for p, v in zip(net._flat_params, views):
            p.grad = v
        net._flat_grad = g
        _b.TrainBuffers.release(buf)


class _DirectBackward:
    """`loss.backward` of the training fast path's loss (an instance attribute of that ONE tensor: anything derived from it --
    `loss * 2`, a sum with another loss -- is an ordinary tensor and differentiates through the engine).  The reference loop's
    own `loss.backward()` (so3_train.py:71: no arguments, gradients cleared) runs the one launch the engine would have run,
    without the engine's hand-over to its device thread (~130 us of host time on a 170-us step); every other call goes to
    torch.Tensor.backward."""
    __slots__ = ("step", "loss")

    def __init__(self, loss, step):
        # The loss is held WEAKLY: a strong reference would close a cycle through the tensor's __dict__ that keeps the step's
        # multi-MB buffers alive for every loss nobody differentiates (a validation pass with gradients enabled) until the cycle
        # collector runs.  `proc(x).backward()` -- the temporary dies when the attribute has been fetched, before the call -- still
        # works: the direct launch needs the step only, and the step is held strongly here.
        self.step, self.loss = step, weakref.ref(loss)

    def __call__(self, gradient=None, retain_graph=None, create_graph=False, inputs=None):
        st, loss = self.step, self.loss()
        if loss is not None:
            loss.__dict__.pop("backward", None)
        if gradient is None and not retain_graph and not create_graph and inputs is None and st.direct_ok():
            st.run_direct()
            return None
        if loss is not None:
            return torch.Tensor.backward(loss, gradient, retain_graph, create_graph, inputs)
        # the loss was a temporary (`proc(x).backward()`) and is gone: what the engine would do with a plain backward() into
        # existing or partly frozen gradients is simple enough to do here; anything else needs the tensor
        if gradient is None and not retain_graph and not create_graph and inputs is None and st.accumulate_ok():
            st.run_accumulate()
            return None
        raise RuntimeError("so3x: backward(...) with arguments, tensor hooks or anomaly mode needs the loss tensor alive: bind it to a "
                           "name first (loss = proc(x); loss.backward(...))")


class _FusedSkewvecLoss(torch.autograd.Function):
    """p_losses of loss_type="skewvec" for the 65-wide RotPredict with bf16 operands (reference diffusion.py:348-357).
    forward = so3x_train_fused: noise draw, q_sample, target, network forward, MSE AND the whole backward down to per-workgroup
    dW partial slabs in ONE kernel (nothing in the math needs the loss before the backward: d loss / d out is per sample);
    backward = so3x_train_bwd_reduce: the fixed-order sum of the slabs times the upstream gradient -> the FLAT parameter
    gradient, handed to the nn.Linear parameters as views (no copy; so3x.flat).  Only the parameters carry gradients
    (SURVEY.md 3.1).  `proc.train_step_kernel = "staged"` selects round 3's form instead (so3x_train_fwd / so3x_train_bwd: three
    kernels with the pre-activations parked in HBM)."""

    @staticmethod
    def forward(ctx, net, proc, x_start, t, axes, unif, *params):
        flat = net._flat
        trap_q, _ = proc._tables()
        dev_rng = proc.rng_counter is not None and (axes is None or t is None)
        kw = dict(quirk_col0=proc.quirk_col0, axes=axes, unif=unif, seed=_rng.seed(),
                  rng_offset=0 if (dev_rng or (axes is not None and t is not None)) else _rng.next_offset(),
                  rng_counter=proc.rng_counter if dev_rng else None, index_base=proc.index_base, guide_q=proc._guide_q)
        st = ctx.step = _FusedStep()
        st.net, st.T, st.n_params, st.carry, st.buf = net, proc.num_timesteps, flat.numel(), None, None
        st.staged = proc.train_step_kernel == "staged"
        if st.staged:
            loss, st.carry, _ = _b.train_fwd(flat, proc._sched, trap_q, x_start, t, **kw)
            return loss
        buf = _b.TrainBuffers.acquire(x_start.numel() // 9, proc.num_timesteps, x_start.device)
        _b.train_fused(buf, flat, proc._sched, trap_q, x_start, t, **kw)
        st.buf = buf
        return buf.loss[0]

    @staticmethod
    def backward(ctx, g):
        st = ctx.step
        if st.staged:
            if st.carry is None:
                raise RuntimeError("so3x: this training step's buffers were already consumed by a backward pass")
            grad = _b.train_bwd(st.carry, st.n_params, st.T, gscale=g)
            st.carry = None
        else:
            if st.buf is None:
                raise RuntimeError("so3x: this training step's buffers were already consumed by a backward pass")
            grad = _b.train_bwd_reduce(st.buf, gscale=g)
            _b.TrainBuffers.release(st.buf)
            st.buf = None
        net = st.net
        if torch.cuda.is_current_stream_capturing() and all(
                p.grad is None and not p._backward_hooks and not p._post_accumulate_grad_hooks for p in net._flat_params if p.requires_grad):
            # inside a graph capture (so3x.graphs.TrainStepGraph's generic form: `loss.backward()` of the captured step, nothing to
            # accumulate into) the flat gradient is hung on the parameters here, without the engine's AccumulateGrad nodes:
            # hipStreamEndCapture does not survive them on this stack (segmentation fault in capture_end, ROCm 7.2 / torch 2.10)
            net._install_flat_grad(grad)
            return (None,) * (6 + len(net._flat_params))
        # Per-parameter views of the flat gradient, whoever asked (backward(), backward(inputs=...), autograd.grad): the engine
        # decides what becomes a .grad.  With nothing to accumulate into it installs the views as they are (no copy), and
        # flat_grad() then recognises them as one flat tensor; the plain `loss.backward()` of the reference loop does not come
        # through here at all (_DirectBackward).
        return (None,) * 6 + tuple(st.net._grad_views(grad))


def _fused_loss(net, proc, x_start, t, axes, unif):
    net._ensure_flat()
    loss = _FusedSkewvecLoss.apply(net, proc, x_start, t, axes, unif, *net._flat_params)
    step = loss.grad_fn.step if loss.grad_fn is not None else None
    if step is None or step.staged:
        return loss
    loss.backward = _DirectBackward(loss, step)
    return loss


class SO3Diffusion(nn.Module):
    def __init__(self, denoise_fn, timesteps=1000, loss_type="skewvec", betas=None, quirk_col0=True):
        super().__init__()
        self.denoise_fn = denoise_fn
        if betas is not None:
            betas = betas.detach().cpu().numpy() if isinstance(betas, torch.Tensor) else np.asarray(betas)
        else:
            betas = cosine_beta_schedule(timesteps)
        betas = np.ascontiguousarray(betas, np.float64)
        (timesteps,) = betas.shape
        self.num_timesteps = int(timesteps)
        self.loss_type = loss_type
        if loss_type not in ("skewvec", "prevstep"):
            # the reference builds this error without raising it (diffusion.py:367) and then fails on an unbound `loss`
            raise ValueError(f"Unexpected loss_type: {loss_type}")
        self.quirk_col0 = quirk_col0
        self.index_base = 0  # global index of this process's first sample (data-parallel shards)
        # None: the Philox offset of each p_losses call is a host counter (so3x.rng).  A device int64 tensor (see
        # so3x.graphs.TrainStepGraph): the offset is read from it by the kernel and incremented on the device after the
        # call, which is what a captured hipGraph of the training step needs to draw fresh noise on every replay.
        self.rng_counter = None
        self.draw_t_in_kernel = True  # forward(): timesteps from the samples' Philox blocks on the training fast path
        self.train_step_kernel = "fused"  # the fast path's device work: "fused" = ONE kernel (so3x_train_fused), "staged" = round 3's three

        sched = _b.schedule_from_betas(betas)  # float64 math, fp32 storage, as diffusion.py:62-92
        for i, name in enumerate(_SCHED_NAMES):
            self.register_buffer(name, torch.from_numpy(sched[i].copy()))
        self.register_buffer("identity", torch.eye(3))
        # kernel-side views: the packed [13, T] table and the two CDF-row tables (built lazily on the GPU)
        self.register_buffer("_sched", torch.from_numpy(sched.copy()), persistent=False)
        self._trap_q = None  # rows for eps_t = sqrt(1 - abar_t)          (p_losses / q_sample)
        self._trap_p = None  # rows for sigma_t = exp(0.5 * logvar_t)     (p_sample)
        self._guide_q = None  # search guide of the q rows (looked up per sample: t differs across the batch)
        self._guide_p = None  # and of the p rows (saves ~7 of the 10 bisection rounds of every reverse step)
        from .flat import PreparedCache
        self._prep = PreparedCache()   # the reverse-chain kernel's prepared state for the current parameters

    # ------------------------------------------------------------------ tables
    def _tables(self):
        dev = self._sched.device
        if self._trap_q is None or self._trap_q.device != dev:
            self._trap_q = _b.igso3_build_tables(self._sched[4])
            self._trap_p = _b.igso3_build_tables(self._sched[12])
            self._guide_q = _b.igso3_build_guide(self._trap_q)
            self._guide_p = _b.igso3_build_guide(self._trap_p)
        return self._trap_q, self._trap_p

    def _prepared(self, net, check=False):
        """so3x_p_sample_prepare's workspace for the 65-wide network's CURRENT parameters (weight image, per-timestep rows, CDF
        records for all T steps): built on first use and whenever the flat parameter buffer, the parameters' tensor versions, the
        out-of-band update epoch (graph replays), the precision or the tables change -- and, for writes torch's versions do not see
        (`p.data`, EMA), when the buffer's fingerprint no longer matches (so3x.flat.PreparedCache: compared at chain starts, after
        idle gaps, on train() / eval() switches and every 256 calls).  A p_sample call then is one kernel launch."""
        from .flat import PARAM_EPOCH
        _, trap_p = self._tables()
        flat = net.flat_params_nograd()
        prec = getattr(net, "chain_precision_code", net.precision_code)
        # (the nn.Parameters share the flat buffer's storage through `.data`, not its version counter: an in-place torch update
        #  -- load_state_dict, torch.optim -- bumps THEIR versions)
        key = (flat.data_ptr(), tuple(p._version for p in net._flat_params), PARAM_EPOCH[0], prec, trap_p.data_ptr(),
               self._guide_p.data_ptr(), flat.device)
        wide = getattr(net, "kind", "") == "resnet255"
        ws = self._prep.get(key, flat, lambda: _b.resnet_p_sample_prepare(flat, self.num_timesteps, prec) if wide
                            else _b.p_sample_prepare(flat, self._sched, trap_p, prec, guide_p=self._guide_p), check=check)
        return ws, prec

    def invalidate_sampling_cache(self):
        """for callers that rewrite the parameters behind torch's back (`p.data` writes, raw pointers, their own captured graphs)
        and want the next p_sample to see it unconditionally"""
        self._prep.invalidate()

    def train(self, mode=True):
        self._prep.check_next()      # a mode switch usually brackets a weight update: the next p_sample compares fingerprints
        return super().train(mode)

    def _fused_net(self, sampling=False):
        """the denoiser when it is one of the two score networks with fused kernels (so3_train / so3_lock_train RotPredict)"""
        from .so3_lock_train import RotPredict as WideRotPredict
        net = self.denoise_fn if isinstance(self.denoise_fn, (RotPredict, WideRotPredict)) else None
        if sampling and net is not None and net.out_type != "skewvec":
            # predict_start_from_noise scales the network output as a [B, 3] skew vector (reference diffusion.py:293-294);
            # with a rotation-matrix head the reference fails there on a shape mismatch
            raise ValueError("so3x: reverse sampling needs a denoiser with out_type='skewvec'")
        return net

    @staticmethod
    def _chain_fn(net):
        return _b.resnet_p_sample_chain if getattr(net, "kind", "") == "resnet255" else _b.p_sample_chain

    @staticmethod
    def _shared_t(t):
        """p_sample draws its noise from model_stdev[0] and skips it when (t == 0).all() (reference diffusion.py:320-325).
        Returns (t[0], every entry equals t[0]); a tensor t costs one host read."""
        if isinstance(t, int):
            return t, True
        tf = t.reshape(-1)
        if tf.numel() == 1:
            return int(tf.item()), True
        t0, same = torch.stack((tf[0], (tf == tf[0]).all().to(tf.dtype))).tolist()
        return int(t0), bool(same)

    # ------------------------------------------------------------------ reference API
    def q_mean_variance(self, x_start, t):
        from .util import so3_lerp
        mean = so3_lerp(self.identity, x_start, self.sqrt_alphas_cumprod[t])
        variance = extract(1.0 - self.alphas_cumprod, t, x_start.shape)
        log_variance = extract(self.log_one_minus_alphas_cumprod, t, x_start.shape)
        return mean, variance, log_variance

    def predict_start_from_noise(self, x_t, t, noise):
        """so3_scale(x_t, sqrt(1/abar)) @ exp(hat(noise * sqrt(1/abar - 1)))^T  (reference diffusion.py:291-297); t per
        sample ([B]) or shared ([1]), gathered on the device as the reference's extract() does"""
        x0hat, _ = _b.p_mean(self._sched, x_t, noise, t, want_x0hat=True)
        return x0hat

    def q_posterior(self, x_start, x_t, t):
        c_1 = _b.so3_scale(x_start, self.posterior_mean_coef1[t])
        c_2 = _b.so3_scale(x_t, self.posterior_mean_coef2[t])
        posterior_mean = _b.rmul(c_1, c_2)
        return posterior_mean, extract(self.posterior_variance, t, t.shape), \
            extract(self.posterior_log_variance_clipped, t, t.shape)

    def p_mean_variance(self, x, t, clip_denoised: bool = False):
        predict = self.denoise_fn(x, t)
        _, model_mean = _b.p_mean(self._sched, x, predict, t)
        return model_mean, extract(self.posterior_variance, t, t.shape), \
            extract(self.posterior_log_variance_clipped, t, t.shape)

    @torch.no_grad()
    def p_sample(self, x, t, clip_denoised=False, repeat_noise=False, axes=None, unif=None):
        """One reverse step (reference diffusion.py:315-326).  t: int64 tensor [B] or [1], or an int.  As in the reference
        the noise scale is model_stdev[0] (t[0]'s) and the noise is skipped only when every t is 0; with all entries
        equal (the only way the reference's own loops call it) the step is ONE fused launch, with mixed entries the mean
        uses each sample's own coefficients (extract(), diffusion.py:291-306)."""
        net = self._fused_net(sampling=True)
        _, trap_p = self._tables()
        wide = net is not None and getattr(net, "kind", "") == "resnet255"
        small = net is not None      # (either fused score network: one launch from its prepared state)
        if small and isinstance(t, torch.Tensor) and t.numel() == 1 and t.is_cuda and t.dtype == torch.int64:
            # the (1,)-shaped t of the reference's own loop (so3_test.py:31): read by the kernel on the device -- no host copy of
            # t, no synchronisation, ONE launch from the prepared state (the reference synchronises here, diffusion.py:320)
            ws, prec = self._prepared(net)
            off = _rng.next_offset(self.num_timesteps) if axes is None else 0
            return _b.p_sample_prepared(ws, self._sched, trap_p, x, 0, 1, t_dev=t, axes=axes, unif=unif, seed=_rng.seed(), rng_offset=off,
                                        index_base=self.index_base, precision=prec, guide_p=self._guide_p, wide=wide)
        t0, same = self._shared_t(t)
        off = _rng.next_offset(self.num_timesteps) if axes is None else 0
        if small and same:
            ws, prec = self._prepared(net, check=t0 == self.num_timesteps - 1)   # a chain's first step: compare fingerprints
            return _b.p_sample_prepared(ws, self._sched, trap_p, x, t0, 1, axes=axes, unif=unif, seed=_rng.seed(), rng_offset=off,
                                        index_base=self.index_base, precision=prec, guide_p=self._guide_p, wide=wide)
        tt = t if isinstance(t, torch.Tensor) else torch.full((1,), t0, device=x.device, dtype=torch.long)
        predict = self.denoise_fn(x, tt)
        _, mean = _b.p_mean(self._sched, x, predict, tt if not same else t0)
        if same and t0 == 0:
            return mean
        n = x.numel() // 9
        smp, _, _ = _b.igso3_sample(trap_p, n, row_const=t0, axes=axes, unif=unif, seed=_rng.seed(), rng_offset=off + t0,
                                    index_base=self.index_base, guide=self._guide_p)
        return _b.rmul(mean, smp.reshape(x.shape))

    @torch.no_grad()
    def p_sample_loop(self, shape, x_init=None):
        """Full reverse chain (reference diffusion.py:328-337).  Initial state IGSO3(eps=1), as the
        reference (its comment says Haar; the code is IGSO3(1))."""
        device = self.betas.device
        b = shape[0]
        if x_init is None:
            x = IsotropicGaussianSO3(eps=torch.ones([], device=device)).sample(shape, index_base=self.index_base)
        else:
            x = x_init
        T = self.num_timesteps
        net = self._fused_net(sampling=True)
        if net is not None:
            _, trap_p = self._tables()
            off = _rng.next_offset(T)
            return self._chain_fn(net)(net.flat_params_nograd(), self._sched, trap_p, x, T - 1, T, seed=_rng.seed(),
                                     rng_offset=off, index_base=self.index_base, precision=getattr(net, "chain_precision_code", net.precision_code),
                                     guide_p=self._guide_p)
        for i in reversed(range(T)):
            x = self.p_sample(x, torch.full((b,), i, device=device, dtype=torch.long))
        return x

    def q_sample(self, x_start, t, noise=None, axes=None, unif=None):
        """so3_scale(x_start, sqrt(abar_t)) @ noise  (reference diffusion.py:339-346)."""
        trap_q, _ = self._tables()
        x_t, _, _ = _b.q_sample_target(self._sched, trap_q, x_start, t, quirk_col0=self.quirk_col0, noise=noise,
                                       axes=axes, unif=unif, seed=_rng.seed(),
                                       rng_offset=_rng.next_offset() if (noise is None and axes is None) else 0,
                                       index_base=self.index_base, want_target=False, guide_q=self._guide_q)
        return x_t

    def p_losses(self, x_start, t, noise=None, axes=None, unif=None):
        """loss_type "skewvec": MSE between the network output and vee(log noise)/eps_t; "prevstep": squared geodesic
        distance between the network's rotation and the step from x_noisy to the posterior mean of the previous timestep
        (reference diffusion.py:348-369)."""
        trap_q, _ = self._tables()
        net = self._fused_net()
        if x_start.numel() > 0 and self._lean(noise):
            return _fused_loss(net, self, x_start, t, axes, unif)  # the training step's fast path
        dev_rng = self.rng_counter is not None and noise is None and axes is None
        prevstep = self.loss_type == "prevstep"
        x_noisy, target, _ = _b.q_sample_target(self._sched, trap_q, x_start, t, quirk_col0=self.quirk_col0, noise=noise,
                                                axes=axes, unif=unif, seed=_rng.seed(),
                                                rng_offset=0 if (dev_rng or noise is not None or axes is not None) else _rng.next_offset(),
                                                index_base=self.index_base, guide_q=self._guide_q,
                                                rng_offset_dev=self.rng_counter if dev_rng else None,
                                                want_target=not prevstep)
        if dev_rng:
            self.rng_counter += 1
        # every t of this process is < num_timesteps: let the fused network gather per-timestep table rows
        if prevstep and net is not None and net.out_type == "rotmat":
            # the fused networks hand over their raw 6 outputs: six2rmat, posterior mean, step = x_noisy^T @ mean,
            # rmat_dist(x_recon, step)^2 and the gradient back through six2rmat are one kernel
            out6 = net(x_noisy, t, t_table=self.num_timesteps, raw=True)
            return _b.prevstep_loss6(self._sched, out6, x_start, x_noisy, t)
        x_recon = net(x_noisy, t, t_table=self.num_timesteps) if net is not None else self.denoise_fn(x_noisy, t)
        if prevstep:
            # any other denoiser returns rotations: posterior mean, step, distance^2 and its gradient wrt x_recon in one kernel
            return _b.prevstep_loss(self._sched, x_recon, x_start, x_noisy, t)
        return _b.mse_loss(x_recon, target)

    def _lean(self, noise):
        """the training step's fast path applies: 65-wide skew-vector RotPredict with bf16 operands under the skewvec loss"""
        net = self._fused_net()
        return (self.loss_type == "skewvec" and noise is None and type(self) is SO3Diffusion and isinstance(net, RotPredict)
                and net.out_type == "skewvec" and net.precision == "bf16" and torch.is_grad_enabled()
                and any(p.requires_grad for p in (net._flat_params or net.net.parameters())))

    def forward(self, x, *args, **kwargs):
        """loss for a batch of rotations at random timesteps (reference diffusion.py:371-374).  On the training fast path
        the timesteps are drawn inside the noising kernel -- uniform on {0..T-1} as the reference's randint, but keyed by
        (seed, global sample index, offset) like the noise, so a sharded run draws what the single-process run draws and a
        replayed hipGraph what the eager loop draws; set `draw_t_in_kernel = False` for torch.randint."""
        b = x.shape[0]
        if self.draw_t_in_kernel and b > 0 and not args and self._lean(kwargs.get("noise")) and set(kwargs) <= {"axes", "unif", "noise"}:
            return _fused_loss(self.denoise_fn, self, x, None, kwargs.get("axes"), kwargs.get("unif"))
        t = torch.randint(0, self.num_timesteps, (b,), device=x.device).long()
        return self.p_losses(x, t, *args, **kwargs)


class ProjectedSO3Diffusion(SO3Diffusion):
    """SO3Diffusion whose denoiser sees `projection(x)` instead of the rotation itself (reference diffusion.py:377-429;
    the projection is any callable, e.g. models.PointCloudProj).  Noising, targets, posterior mean and noise are the fused
    kernels of the base class; the denoiser and the projection are the caller's."""

    def p_mean_variance(self, x, t, clip_denoised: bool = False):
        predict = self.denoise_fn(self.projection(x), t)
        _, model_mean = _b.p_mean(self._sched, x, predict, t)
        return model_mean, extract(self.posterior_variance, t, t.shape), \
            extract(self.posterior_log_variance_clipped, t, t.shape)

    @torch.no_grad()
    def p_sample(self, x, t, clip_denoised=False, repeat_noise=False, axes=None, unif=None):
        t0, same = self._shared_t(t)
        mean, _, _ = self.p_mean_variance(x, t if isinstance(t, torch.Tensor) else torch.full((1,), t0, device=x.device,
                                                                                            dtype=torch.long))
        if same and t0 == 0:
            return mean
        _, trap_p = self._tables()
        off = _rng.next_offset(self.num_timesteps) if axes is None else 0
        smp, _, _ = _b.igso3_sample(trap_p, x.numel() // 9, row_const=t0, axes=axes, unif=unif, seed=_rng.seed(),
                                    rng_offset=off + t0, index_base=self.index_base, guide=self._guide_p)
        return _b.rmul(mean, smp.reshape(x.shape))

    @torch.no_grad()
    def p_sample_loop(self, shape, projection, x_init=None):
        """reference diffusion.py:391-401: starts from the Q factor of a Gaussian matrix (which may have det -1, as there)"""
        self.projection = projection
        device = self.betas.device
        b = shape[0]
        x = torch.linalg.qr(torch.randn((b, 3, 3), device=device))[0] if x_init is None else x_init
        for i in reversed(range(self.num_timesteps)):
            x = self.p_sample(x, torch.full((b,), i, device=device, dtype=torch.long))
        return x

    def p_losses(self, x_start, t, noise=None, axes=None, unif=None):
        trap_q, _ = self._tables()
        x_noisy, target, _ = _b.q_sample_target(self._sched, trap_q, x_start, t, quirk_col0=self.quirk_col0, noise=noise,
                                                axes=axes, unif=unif, seed=_rng.seed(),
                                                rng_offset=_rng.next_offset() if (noise is None and axes is None) else 0,
                                                index_base=self.index_base, guide_q=self._guide_q)
        x_recon = self.denoise_fn(self.projection(x_noisy), t)
        return _b.mse_loss(x_recon, target)

    def forward(self, x, projection, *args, **kwargs):
        self.projection = projection
        b = x.shape[0]
        t = torch.randint(0, self.num_timesteps, (b,), device=x.device).long()
        return self.p_losses(x, t, *args, **kwargs)
