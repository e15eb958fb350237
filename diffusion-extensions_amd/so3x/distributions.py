"""IsotropicGaussianSO3 with the reference's interface (reference distributions.py:8-81).

eps may be a 0-d tensor (one CDF row, the p_sample case) or a batch (one row per
element, the p_losses case).  The CDF rows are built on the GPU at construction
(fp64 density, fp32 trapezoid, double-accumulated prefix sum -- the reference's mixed
precision).  Sampling uses explicit draws when given (parity tests) or in-kernel Philox."""
import torch

from . import backend as _b
from . import rng as _rng

__all__ = ["IsotropicGaussianSO3", "Bingham"]


class _LogProb(torch.autograd.Function):
    """log_prob with the gradient wrt the rotations the reference gets from torch autograd (distributions.py:74-77 and its
    use at 189-190): the dense d logp / dR comes out of the same kernel launch as the value."""

    @staticmethod
    def forward(ctx, rotations, eps):
        logp, _, grad = _b.igso3_logprob_score(rotations, eps, want_score=False, want_grad=True)
        ctx.save_for_backward(grad)
        return logp

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return grad * g[..., None], None


class IsotropicGaussianSO3:
    def __init__(self, eps: torch.Tensor, mean: torch.Tensor = None, quirk_col0: bool = True):
        if not isinstance(eps, torch.Tensor):
            raise TypeError("eps must be a tensor on the MI355X device")
        self.eps = eps
        self._mean = mean.to(eps) if mean is not None else None  # None == identity (reference default eye(3))
        # reference distributions.py:42-43: with batched eps the interpolation weight is gathered from
        # column 0 of the table for every sample; reproduced by default, quirk_col0=False fixes it.
        self.quirk_col0 = quirk_col0
        self.trap = _b.igso3_build_tables(eps.reshape(-1))  # [numel(eps), 999]; reference keeps [999, *eps.shape]
        self._guide = None  # search guide of the rows (bit-identical angles, 2-3 probes instead of 10), built on first sample()

    @property
    def mean(self):
        if self._mean is None:
            return torch.eye(3, device=self.eps.device)
        return self._mean

    def sample(self, sample_shape=torch.Size(), axes=None, unif=None, index_base=0):
        sample_shape = tuple(sample_shape)
        eshape = tuple(self.eps.shape)
        n_eps = self.eps.numel()
        n_rep = 1
        for s in sample_shape:
            n_rep *= s
        n = n_rep * n_eps
        batched = self.eps.dim() > 0
        row_idx = None
        if batched:
            row_idx = torch.arange(n_eps, device=self.eps.device).repeat(n_rep) if n_rep > 1 else \
                torch.arange(n_eps, device=self.eps.device)
        if self._guide is None:
            self._guide = _b.igso3_build_guide(self.trap)
        out, _, _ = _b.igso3_sample(self.trap, n, row_idx=row_idx, row_const=0, guide=self._guide,
                                    quirk_col0=bool(batched and self.quirk_col0), axes=axes, unif=unif,
                                    seed=_rng.seed(), rng_offset=_rng.next_offset() if axes is None else 0,
                                    index_base=index_base, mean=self._mean)
        return out.reshape(sample_shape + eshape + (3, 3))

    def _eps_ft(self, t: torch.Tensor) -> torch.Tensor:
        """Closed-form density wrt the Haar measure at angle(s) t (reference distributions.py:53-72)."""
        if self.eps.dim() == 0 or self.eps.numel() == 1:
            return _b.igso3_eps_ft(t.contiguous(), self.eps.reshape(1))
        tb, eb = torch.broadcast_tensors(t, self.eps)
        return _b.igso3_eps_ft(tb.contiguous(), eb.contiguous())

    def log_prob(self, rotations: torch.Tensor) -> torch.Tensor:
        """log f(angle(R)), shape [..., 1] (reference distributions.py:74-77)."""
        if torch.is_grad_enabled() and rotations.requires_grad:
            return _LogProb.apply(rotations, self.eps)  # differentiable wrt the rotations, as under the reference's autograd
        logp, _, _ = _b.igso3_logprob_score(rotations, self.eps, want_score=False)
        return logp

    def log_prob_and_score(self, rotations: torch.Tensor, dense_grad: bool = False):
        """log-prob plus its gradient: tangent 3-vector (f'/f * axis), or the autograd-shaped
        d logp / dR [..., 3, 3] that the reference obtains with torch.autograd.grad
        (reference distributions.py:189-190)."""
        logp, score, grad = _b.igso3_logprob_score(rotations, self.eps, want_score=not dense_grad, want_grad=dense_grad)
        return logp, (grad if dense_grad else score)


class Bingham(torch.distributions.MultivariateNormal):
    """Antipodally symmetric distribution on unit quaternions: a zero-mean Gaussian 4-vector, normalised
    (reference distributions.py:113-127).  It is the DATA source of bingham_train / bingham_test (samples go through
    quat_to_rmat), not part of the diffusion hot path: plain torch on whatever device `loc` lives on."""

    def __init__(self, loc, covariance_matrix=None, precision_matrix=None, scale_tril=None, validate_args=None):
        super().__init__(torch.zeros_like(loc), covariance_matrix, precision_matrix, scale_tril, validate_args)  # location is always 0

    def rsample(self, sample_shape=torch.Size()):
        vals = super().rsample(sample_shape)
        return vals / vals.norm(dim=-1, keepdim=True)
