"""SE(3) = SO(3) x R^3 layer with the reference's names:
AffineT / AffineGrad / se3_scale / se3_lerp (reference util.py:10-56, 364-385),
IGSO3xR3 (distributions.py:84-110), SE3Diffusion (diffusion.py:432-522) and
move_prot (prot_util.py:73-81).  Rotations run on the SO(3) kernels, the fused
SE(3) steps on so3x_se3.hip; shifts of the small helper ops are single torch
element-wise expressions."""
from collections import namedtuple

import numpy as np
import torch
import torch.nn as nn

from . import backend as _b
from . import rng as _rng
from .diffusion import _SCHED_NAMES, cosine_beta_schedule, extract
from .distributions import IsotropicGaussianSO3

__all__ = ["AffineT", "AffineGrad", "ProtData", "se3_scale", "se3_lerp", "IGSO3xR3", "SE3Diffusion", "ProjectedSE3Diffusion",
           "move_prot", "move_prots", "ProtProjection"]

ProtData = namedtuple("ProtData", ["residues", "positions", "angles"])


class AffineT(object):
    """Rigid transform: rot [..., 3, 3] and shift [..., 3] (reference util.py:10-42)."""

    def __init__(self, rot: torch.Tensor, shift: torch.Tensor):
        self.rot = rot
        self.shift = shift

    def __len__(self):
        return max(len(self.rot), len(self.shift))

    def __getitem__(self, item):
        return AffineT(self.rot[item], self.shift[item])

    @property
    def device(self):
        return self.rot.device

    @property
    def shape(self):
        return self.shift.shape

    def to(self, device):
        return AffineT(self.rot.to(device), self.shift.to(device))

    def detach(self):
        return AffineT(self.rot.detach(), self.shift.detach())

    @classmethod
    def from_euler(cls, euls: torch.Tensor, shift: torch.Tensor):
        """reference util.py:35-38"""
        from .util import euler_to_rmat
        return cls(euler_to_rmat(*torch.unbind(euls, dim=-1)), shift)


class AffineGrad(object):
    """Denoiser output: rot_g [..., 3] (tangent vector) and shift_g [..., 3] (reference util.py:45-56)."""

    def __init__(self, rot_g, shift_g):
        self.rot_g = rot_g
        self.shift_g = shift_g

    def __len__(self):
        return max(len(self.rot_g), len(self.shift_g))

    def __getitem__(self, item):
        return AffineGrad(self.rot_g[item], self.shift_g[item])


def se3_scale(transf: AffineT, scalars) -> AffineT:
    """reference util.py:382-385"""
    return AffineT(_b.so3_scale(transf.rot, scalars), transf.shift * scalars[..., None])


def se3_lerp(transf_a: AffineT, transf_b: AffineT, weight: torch.Tensor) -> AffineT:
    """reference util.py:364-379"""
    return AffineT(_b.so3_lerp(transf_a.rot, transf_b.rot, weight), torch.lerp(transf_a.shift, transf_b.shift, weight))


def move_prot(transf: AffineT, protein: ProtData) -> ProtData:
    """Rigid move of residue positions [.., L, 3] and frames [.., L, 3, 3] about the residue centroid
    (reference prot_util.py:73-81); batched over leading structure dimensions."""
    pos, fr = protein.positions, protein.angles
    L = pos.shape[-2]
    rot = transf.rot.reshape(-1, 3, 3)
    out_pos, out_fr = _b.rigid_move(rot, transf.shift.reshape(-1, 3), pos.reshape(rot.shape[0], L, 3),
                                    fr.reshape(rot.shape[0], L, 3, 3) if fr is not None else None)
    return ProtData(protein.residues, out_pos.reshape(pos.shape), out_fr.reshape(fr.shape) if fr is not None else None)


def move_prots(transf: AffineT, proteins) -> list:
    """Move a collection of structures about their SHARED centroid (reference prot_util.py:61-70): one transform, several
    proteins of possibly different lengths.  The structures are concatenated along the residue axis, moved by the rigid-move
    kernel in one launch (its centroid is then the shared one), and split again."""
    proteins = list(proteins)
    lens = [p.positions.shape[-2] for p in proteins]
    both = ProtData(None, torch.cat([p.positions for p in proteins], dim=-2), torch.cat([p.angles for p in proteins], dim=-3))
    moved = move_prot(transf, both)
    pos = torch.split(moved.positions, lens, dim=-2)
    ang = torch.split(moved.angles, lens, dim=-3)
    return [ProtData(p.residues, a, b) for p, a, b in zip(proteins, pos, ang)]


class ProtProjection(nn.Module):
    """Projection of ProjectedSE3Diffusion for docking data (reference prot_util.py:102-117): `data` is a sequence of
    (receptor, ligand) ProtData pairs; the i-th transform moves the i-th ligand, the receptor stays.  se3=False takes
    [n, 6] = (Euler angles, shift) instead of an AffineT.
    `data` may also be a so3x.backend.ProtBatch (the same pairs already concatenated with offsets, what a loader builds once per
    batch): then all ligands move in ONE launch (so3x_rigid_move_ragged) and the result is a ProtBatch again -- the form
    ProtNet takes without re-concatenating ~6 B small tensors per call."""

    def __init__(self, data, se3=True):
        super().__init__()
        self.data = data
        self.se3 = se3

    def forward(self, transforms):
        if self.se3:
            tfs = transforms
        else:
            from .util import euler_to_rmat
            tfs = AffineT(euler_to_rmat(*torch.unbind(transforms[..., :3], -1)), transforms[..., 3:])
        if isinstance(self.data, _b.ProtBatch):
            n = len(self.data)
            pos, ang = _b.rigid_move_ragged(tfs.rot[:n], tfs.shift[:n], self.data.lig[1], self.data.lig[2], self.data.lig_off)
            return self.data.with_ligands(pos, ang)
        return [(pair[0], move_prot(tfs[i], pair[1])) for i, pair in enumerate(self.data)]


class IGSO3xR3:
    """IGSO(3) x isotropic Gaussian on R^3 (reference distributions.py:84-110)."""

    def __init__(self, eps: torch.Tensor, mean: AffineT = None, shift_scale=1.0):
        self.eps = eps
        self.shift_scale = shift_scale
        self._mean = mean
        self.igso3 = IsotropicGaussianSO3(eps=eps, mean=mean.rot if (mean is not None and mean.rot.dim() == 2) else None)

    @property
    def mean(self):
        return self._mean

    def sample(self, sample_shape=torch.Size(), axes=None, unif=None, znorm=None):
        rot = self.igso3.sample(sample_shape, axes=axes, unif=unif)
        if self._mean is not None and self._mean.rot.dim() > 2:
            rot = _b.rmul(self._mean.rot, rot)  # per-sample mean rotations
        shape = tuple(sample_shape) + tuple(self.eps.shape) + (3,)
        if znorm is None:
            znorm = torch.randn(shape, device=self.eps.device)
        shift = znorm.reshape(shape) * (self.eps[..., None] * self.shift_scale)
        if self._mean is not None:
            shift = shift + self._mean.shift
        return AffineT(rot, shift)

    def log_prob(self, value: AffineT) -> torch.Tensor:
        """log density of an affine transform (reference distributions.py:103-106): the IGSO(3) log-density of the rotation
        ([..., 1], which like the reference's ignores the mean rotation) plus the per-coordinate Normal log-density of the
        shift ([..., 3]); the sum broadcasts to [..., 3] exactly as the reference's does."""
        rot_prob = self.igso3.log_prob(value.rot)
        loc = self._mean.shift if self._mean is not None else torch.zeros_like(value.shift)
        scale = self.eps[..., None] * self.shift_scale
        shift_prob = torch.distributions.Normal(loc=loc, scale=scale).log_prob(value.shift)
        return rot_prob + shift_prob


class SE3Diffusion(nn.Module):
    """reference diffusion.py:432-522.  denoise_fn(AffineT, t) -> AffineGrad (any torch module/callable)."""

    def __init__(self, denoise_fn, timesteps=1000, loss_type="grad_mse", betas=None, shift_scale=75.0, quirk_col0=True,
                 shared_rot_noise=True):
        super().__init__()
        self.denoise_fn = denoise_fn
        if betas is not None:
            betas = betas.detach().cpu().numpy() if isinstance(betas, torch.Tensor) else np.asarray(betas)
        else:
            betas = cosine_beta_schedule(timesteps)
        betas = np.ascontiguousarray(betas, np.float64)
        self.num_timesteps = int(betas.shape[0])
        if loss_type != "grad_mse":
            raise NotImplementedError("so3x: SE3Diffusion implements loss_type='grad_mse' (reference diffusion.py:513)")
        self.loss_type = loss_type
        self.shift_scale = shift_scale
        self.quirk_col0 = quirk_col0
        # reference p_sample draws ONE rotation noise for the whole batch (diffusion.py:482 with scalar eps and an
        # empty sample shape, distributions.py:98-101); kept by default, False = one draw per sample
        self.shared_rot_noise = shared_rot_noise
        self.index_base = 0
        sched = _b.schedule_from_betas(betas)
        for i, name in enumerate(_SCHED_NAMES):
            self.register_buffer(name, torch.from_numpy(sched[i].copy()))
        self.register_buffer("identity", torch.eye(3))
        self.register_buffer("_sched", torch.from_numpy(sched.copy()), persistent=False)
        self._sigma_host = sched[12].copy()
        self._trap_q = None
        self._trap_p = None
        self._guide_q = None

    def _tables(self):
        dev = self._sched.device
        if self._trap_q is None or self._trap_q.device != dev:
            self._trap_q = _b.igso3_build_tables(self._sched[4])
            self._trap_p = _b.igso3_build_tables(self._sched[12])
            self._guide_q = _b.igso3_build_guide(self._trap_q)
        return self._trap_q, self._trap_p

    @staticmethod
    def _shared_t(t):
        """the fused SE(3) mean kernel takes ONE timestep per call (how the reference's own loops call p_sample,
        diffusion.py:485-494); a batch of differing timesteps is refused rather than silently given sample 0's"""
        if isinstance(t, int):
            return t
        tf = t.reshape(-1)
        if tf.numel() == 1:
            return int(tf.item())
        t0, same = torch.stack((tf[0], (tf == tf[0]).all().to(tf.dtype))).tolist()
        if not same:
            raise ValueError("so3x: SE3Diffusion.p_mean_variance / p_sample need one shared timestep per call")
        return int(t0)

    def q_mean_variance(self, x_start, t):
        mean = se3_scale(x_start, self.sqrt_alphas_cumprod[t])
        return mean, extract(1.0 - self.alphas_cumprod, t, x_start.shape), \
            extract(self.log_one_minus_alphas_cumprod, t, x_start.shape)

    def predict_start_from_noise(self, x_t: AffineT, t, noise: AffineGrad) -> AffineT:
        """x_0 estimate from a predicted noise (reference diffusion.py:444-455): rotation = so3_scale(x_t, sqrt(1/abar))
        @ exp(hat(rot_g * sqrt(1/abar - 1)))^T, shift = x_t.shift * sqrt(1/abar) - shift_g * sqrt(1/abar - 1)"""
        k = self.sqrt_recip_alphas_cumprod[t]
        ns = self.sqrt_recipm1_alphas_cumprod[t][..., None]
        x_t_term = se3_scale(x_t, k)
        noise_rot = _b.exp_skewvec((noise.rot_g * ns).contiguous())
        return AffineT(_b.rmul(x_t_term.rot, noise_rot, transpose_b=True), x_t_term.shift - noise.shift_g * ns)

    def q_posterior(self, x_start: AffineT, x_t: AffineT, t):
        """reference diffusion.py:457-464"""
        c_1 = se3_scale(x_start, self.posterior_mean_coef1[t])
        c_2 = se3_scale(x_t, self.posterior_mean_coef2[t])
        return AffineT(_b.rmul(c_1.rot, c_2.rot), c_1.shift + c_2.shift), extract(self.posterior_variance, t, t.shape), \
            extract(self.posterior_log_variance_clipped, t, t.shape)

    def p_mean_variance(self, x: AffineT, t, clip_denoised: bool = False):
        predict = self.denoise_fn(x, t)
        mean_rot, mean_shift = _b.se3_p_mean(self._sched, x.rot, x.shift, predict.rot_g, predict.shift_g, self._shared_t(t))
        return AffineT(mean_rot, mean_shift), extract(self.posterior_variance, t, t.shape), \
            extract(self.posterior_log_variance_clipped, t, t.shape)

    @torch.no_grad()
    def p_sample(self, x: AffineT, t, clip_denoised=False, repeat_noise=False, axes=None, unif=None, znorm=None):
        t0 = self._shared_t(t)
        mean, _, _ = self.p_mean_variance(x, t)
        if t0 == 0:
            return mean
        _, trap_p = self._tables()
        off = _rng.next_offset() if axes is None else 0
        rot, shift = _b.se3_p_noise(trap_p[t0], float(self._sigma_host[t0]), self.shift_scale, mean.rot, mean.shift, axes=axes,
                                    unif=unif, znorm=znorm, seed=_rng.seed(), rng_offset=off, index_base=self.index_base,
                                    shared_rot=self.shared_rot_noise)
        return AffineT(rot, shift)

    @torch.no_grad()
    def p_sample_loop(self, shape, x_init: AffineT = None):
        """Full reverse chain (reference diffusion.py:486-495, which starts from the Q factor of a Gaussian matrix as a bare
        tensor and so cannot run as written; here the start is that rotation with a zero shift, or `x_init`)."""
        device = self.betas.device
        b = shape[0]
        if x_init is None:
            rot, _ = torch.linalg.qr(torch.randn((b, 3, 3), device=device))
            x_init = AffineT(rot.contiguous(), torch.zeros(b, 3, device=device))
        x = x_init
        for i in reversed(range(self.num_timesteps)):
            x = self.p_sample(x, torch.full((b,), i, device=device, dtype=torch.long))
        return x

    def q_sample(self, x_start: AffineT, t, noise: AffineT = None, axes=None, unif=None, znorm=None):
        if noise is not None:
            x_blend = se3_scale(x_start, self.sqrt_alphas_cumprod[t])
            return AffineT(_b.rmul(x_blend.rot, noise.rot), x_blend.shift + noise.shift)
        trap_q, _ = self._tables()
        xt_rot, xt_shift, _, _ = _b.se3_q_sample_target(
            self._sched, trap_q, self.shift_scale, x_start.rot, x_start.shift, t, quirk_col0=self.quirk_col0, axes=axes,
            unif=unif, znorm=znorm, seed=_rng.seed(), rng_offset=_rng.next_offset() if axes is None else 0,
            index_base=self.index_base, want_targets=False, guide_q=self._guide_q)
        return AffineT(xt_rot, xt_shift)

    def p_losses(self, x_start: AffineT, t, noise=None, axes=None, unif=None, znorm=None):
        trap_q, _ = self._tables()
        xt_rot, xt_shift, tg_rot, tg_shift = _b.se3_q_sample_target(
            self._sched, trap_q, self.shift_scale, x_start.rot, x_start.shift, t, quirk_col0=self.quirk_col0, axes=axes,
            unif=unif, znorm=znorm, seed=_rng.seed(), rng_offset=_rng.next_offset() if axes is None else 0,
            index_base=self.index_base, guide_q=self._guide_q)
        x_recon = self.denoise_fn(AffineT(xt_rot, xt_shift), t)
        return _b.mse_loss(x_recon.shift_g, tg_shift) + _b.mse_loss(x_recon.rot_g, tg_rot)

    def forward(self, x: AffineT, *args, **kwargs):
        b = len(x)
        t = torch.randint(0, self.num_timesteps, (b,), device=x.device).long()
        return self.p_losses(x, t, *args, **kwargs)


class ProjectedSE3Diffusion(SE3Diffusion):
    """SE3Diffusion whose denoiser sees `projection(x)` (reference diffusion.py:525-573).  The reference's constructor does
    not forward `shift_scale` to its base class and then sets the attribute itself (527-529); the effect -- the given
    shift_scale is used everywhere -- is what this class does."""

    def p_mean_variance(self, x: AffineT, t, clip_denoised: bool = False):
        predict = self.denoise_fn(self.projection(x), t)
        mean_rot, mean_shift = _b.se3_p_mean(self._sched, x.rot, x.shift, predict.rot_g, predict.shift_g, self._shared_t(t))
        return AffineT(mean_rot, mean_shift), extract(self.posterior_variance, t, t.shape), \
            extract(self.posterior_log_variance_clipped, t, t.shape)

    @torch.no_grad()
    def p_sample_loop(self, shape, projection, x_init: AffineT = None):
        """reference diffusion.py:539-550: Q factor of a Gaussian matrix and a unit Gaussian shift as the initial state"""
        self.projection = projection
        device = self.betas.device
        b = shape[0]
        if x_init is None:
            x_init = AffineT(torch.linalg.qr(torch.randn((b, 3, 3), device=device))[0], torch.randn((b, 3), device=device))
        x = x_init
        for i in reversed(range(self.num_timesteps)):
            x = self.p_sample(x, torch.full((b,), i, device=device, dtype=torch.long))
        return x

    def p_losses(self, x_start: AffineT, t, noise=None, axes=None, unif=None, znorm=None):
        trap_q, _ = self._tables()
        xt_rot, xt_shift, tg_rot, tg_shift = _b.se3_q_sample_target(
            self._sched, trap_q, self.shift_scale, x_start.rot, x_start.shift, t, quirk_col0=self.quirk_col0, axes=axes,
            unif=unif, znorm=znorm, seed=_rng.seed(), rng_offset=_rng.next_offset() if axes is None else 0,
            index_base=self.index_base, guide_q=self._guide_q)
        x_recon = self.denoise_fn(self.projection(AffineT(xt_rot, xt_shift)), t)
        return _b.mse_loss(x_recon.shift_g, tg_shift) + _b.mse_loss(x_recon.rot_g, tg_rot)

    def forward(self, x: AffineT, projection, *args, **kwargs):
        self.projection = projection
        b = len(x)
        t = torch.randint(0, self.num_timesteps, (b,), device=x.device).long()
        return self.p_losses(x, t, *args, **kwargs)
