"""Rotation algebra with the reference's names and argument conventions
(reference util.py:79-107, 164-252, 315-361), computed by the HIP kernels behind
so3x.backend.  Tensors must live on the MI355X ('cuda') device; leading batch
dimensions are free, rotations end in (3, 3), fp32."""
from typing import Tuple

import torch

from . import backend as _b

__all__ = ["skew2vec", "vec2skew", "orthogonalise", "rmat2six", "six2rmat", "log_rmat", "aa_to_rmat", "rmat_to_aa", "quat_to_rmat",
           "rmat_dist", "so3_lerp", "so3_bezier", "so3_scale", "rmat_to_euler", "euler_to_rmat", "to_device", "init_from_dict", "identity", "masked_mean", "cycle", "rmat_cosine_dist", "rmat_gaussian_kernel", "rmat_cosine_kernel",
           "MMD", "Ker_2samp_test", "Ker_2samp_log_prob"]


def skew2vec(skew: torch.Tensor) -> torch.Tensor:
    """vee map (reference util.py:79-84): (S21, -S20, S10).  Pure indexing: stays in torch."""
    return torch.stack((skew[..., 2, 1], -skew[..., 2, 0], skew[..., 1, 0]), dim=-1)


def vec2skew(vec: torch.Tensor) -> torch.Tensor:
    """hat map (reference util.py:87-92)."""
    z = torch.zeros_like(vec[..., 0])
    x, y, w = vec[..., 0], vec[..., 1], vec[..., 2]
    return torch.stack((z, -w, y, w, z, -x, -y, x, z), dim=-1).reshape(vec.shape[:-1] + (3, 3))


def orthogonalise(mat: torch.Tensor) -> torch.Tensor:
    """Reference util.py:95-107: the leading 3x3 block of every matrix is replaced by U round(S) V^T of its SVD (singular
    values snapped to integers); one HIP kernel (so3x_orthogonalise).  The rotations this backend produces come out of
    closed-form Rodrigues formulas and are orthogonal to fp32 rounding, so the kernels that build them skip this step
    (there it is the identity map); called directly it does what the reference's does."""
    if mat.shape[-2:] == (3, 3):
        return _b.orthogonalise(mat)
    out = mat.clone()
    out[..., :3, :3] = _b.orthogonalise(mat[..., :3, :3].contiguous())
    return out


def rmat2six(x: torch.Tensor) -> torch.Tensor:
    """First two rows of a rotation, flattened (reference util.py:60-64).  Pure indexing: stays in torch."""
    return torch.flatten(x[..., :2, :], -2, -1)


def six2rmat(x: torch.Tensor) -> torch.Tensor:
    """Gram-Schmidt of two 3-vectors -> rotation with rows b1, b2, b1 x b2 (reference util.py:67-76); differentiable
    (closed-form backward kernel), it is the head of RotPredict(out_type="rotmat")."""
    return _b.six2rmat(x)


def _needs_grad(*xs):
    return torch.is_grad_enabled() and any(isinstance(x, torch.Tensor) and x.requires_grad for x in xs)


def log_rmat(r_mat: torch.Tensor) -> torch.Tensor:
    """Matrix log as a skew-symmetric matrix (reference util.py:164-192); differentiable like the reference's."""
    return _b.log_rmat_ad(r_mat) if _needs_grad(r_mat) else _b.log_rmat(r_mat)


def aa_to_rmat(rot_axis: torch.Tensor, ang: torch.Tensor) -> torch.Tensor:
    """Axis (any norm) + angle -> rotation matrix (reference util.py:195-205)."""
    return _b.aa_to_rmat(rot_axis, ang)


def rmat_to_aa(r_mat: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """Rotation -> (axis [...,3], angle [...,1] in [0, pi]) (reference util.py:208-219)."""
    return _b.rmat_to_aa(r_mat)


def quat_to_rmat(quaternions: torch.Tensor) -> torch.Tensor:
    """Real-first quaternions of any norm -> rotation matrices (reference util.py:222-252)."""
    return _b.quat_to_rmat(quaternions)


def rmat_dist(input: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    """Geodesic distance ||log(input^T target)||_F (reference util.py:315-322); differentiable like the reference's."""
    if _needs_grad(input, target):
        if input.shape != target.shape:
            input, target = torch.broadcast_tensors(input, target)  # autograd sums the expanded gradient back
        return _b.rmat_dist_ad(input.contiguous(), target.contiguous())
    return _b.rmat_dist(input, target)


def so3_lerp(rot_a: torch.Tensor, rot_b: torch.Tensor, weight: torch.Tensor) -> torch.Tensor:
    """Geodesic interpolation (reference util.py:325-338); rot_a may be a single (3,3)."""
    return _b.so3_lerp(rot_a, rot_b, weight)


def so3_scale(rmat: torch.Tensor, scalars: torch.Tensor) -> torch.Tensor:
    """exp(scalars * log(rmat)) (reference util.py:349-361); scalars has 1 or batch elements."""
    return _b.so3_scale(rmat, scalars)


def rmat_to_euler(rmat: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """(x, y, z) Euler angles of a rotation, the reference's plotting decomposition (util.py:388-393: y from
    atan2(R20, sqrt(R00^2 + R10^2)), i.e. its own sign convention for R_y).  Analysis helper: plain tensor ops."""
    sy = torch.sqrt(rmat[..., 0, 0] * rmat[..., 0, 0] + rmat[..., 1, 0] * rmat[..., 1, 0])
    return (torch.atan2(rmat[..., 2, 1], rmat[..., 2, 2]), torch.atan2(rmat[..., 2, 0], sy),
            torch.atan2(rmat[..., 1, 0], rmat[..., 0, 0]))


def euler_to_rmat(x: torch.Tensor, y: torch.Tensor, z: torch.Tensor) -> torch.Tensor:
    """R_z(z) @ R_y(y) @ R_x(x) with the reference's sign convention for R_y (R_y[2,0] = +sin y, R_y[0,2] = -sin y;
    reference util.py:396-423).  Host-side data preparation (two constant rotations in so3_lock_train.py:76-77):
    plain tensor ops, any device."""
    cx, sx, cy, sy, cz, sz = torch.cos(x), torch.sin(x), torch.cos(y), torch.sin(y), torch.cos(z), torch.sin(z)
    one, zero = torch.ones_like(cx), torch.zeros_like(cx)
    Rx = torch.stack((one, zero, zero, zero, cx, -sx, zero, sx, cx), -1).reshape(*cx.shape, 3, 3)
    Ry = torch.stack((cy, zero, -sy, zero, one, zero, sy, zero, cy), -1).reshape(*cy.shape, 3, 3)
    Rz = torch.stack((cz, -sz, zero, sz, cz, zero, zero, zero, one), -1).reshape(*cz.shape, 3, 3)
    return Rz @ Ry @ Rx


def so3_bezier(*rots: torch.Tensor, weight: torch.Tensor) -> torch.Tensor:
    """De Casteljau curve through rotations by repeated geodesic interpolation (reference util.py:340-346, whose recursion
    passes the tuples un-splatted and cannot run; this is the evidently intended recursion)."""
    if len(rots) < 2:
        raise ValueError("so3_bezier needs at least two rotations")
    if len(rots) == 2:
        return so3_lerp(rots[0], rots[1], weight)
    return so3_lerp(so3_bezier(*rots[:-1], weight=weight), so3_bezier(*rots[1:], weight=weight), weight)


def to_device(device, *objects, non_blocking=False):
    """Move tensors, and (nested) iterables of tensors, to a device (reference util.py:426-437, minus its protein record type)."""
    moved = []
    for obj in objects:
        if isinstance(obj, torch.Tensor):
            moved.append(obj.to(device, non_blocking=non_blocking))
        elif isinstance(obj, (list, tuple)) or hasattr(obj, "__iter__"):
            moved.append(to_device(device, *obj, non_blocking=non_blocking))
        else:
            raise RuntimeError(f"Cannot move object of type {type(obj)} to {device}")
    return moved


def init_from_dict(argdict, *classes):
    """Construct each class from the entries of one dict that its signature names; other entries are ignored
    (reference util.py:440-460)."""
    import inspect
    objs = []
    for cls in classes:
        names = [k for k, v in inspect.signature(cls).parameters.items() if v.kind == inspect.Parameter.POSITIONAL_OR_KEYWORD]
        objs.append(cls(**{k: v for k, v in argdict.items() if k in names}))
    return objs


def identity(x):
    return x


def masked_mean(tensor: torch.Tensor, mask: torch.Tensor, dim=-1) -> torch.Tensor:
    """Mean over `dim` of the entries selected by a boolean mask, 0 where nothing is selected; zeroes the masked-out
    entries of `tensor` IN PLACE, as the reference does (util.py:467-475)."""
    mask = mask[(..., *((None,) * (tensor.dim() - mask.dim())))]
    tensor.masked_fill_(~mask, 0.0)
    count = mask.sum(dim=dim)
    mean = tensor.sum(dim=dim) / count.clamp(min=1.0)
    mean.masked_fill_(count == 0, 0.0)
    return mean


def cycle(iterable):
    """Endless iterator over a DataLoader-like iterable (reference util.py helper)."""
    while True:
        for x in iterable:
            yield x


# ---------------------------------------------------------------------------------------------
# kernel two-sample statistics (reference util.py:110-151, 254-312)
# ---------------------------------------------------------------------------------------------
def rmat_cosine_dist(m1: torch.Tensor, m2: torch.Tensor) -> torch.Tensor:
    """1 - cos(angle(m2^T m1)) (reference util.py:110-125); trace of a 3x3 product: plain indexing/arithmetic."""
    tra = (m2 * m1).sum(dim=(-1, -2))  # tr(m2^T m1) = sum_ij m2_ij m1_ij
    return 1 - (tra - 1) / 2


def rmat_gaussian_kernel(m1: torch.Tensor, m2: torch.Tensor) -> torch.Tensor:
    """exp(-rmat_dist(m1, m2)), broadcasting like the reference (util.py:128-134)."""
    return torch.exp(-rmat_dist(m1, m2))


def rmat_cosine_kernel(m1: torch.Tensor, m2: torch.Tensor) -> torch.Tensor:
    """(tr(m2^T m1) - 1)/2 (reference util.py:136-151)"""
    return ((m2 * m1).sum(dim=(-1, -2)) - 1) / 2


_FUSED_KERNELS = {rmat_gaussian_kernel: _b.KERNEL_GAUSSIAN, rmat_cosine_kernel: _b.KERNEL_COSINE}


def MMD(X: torch.Tensor, Y: torch.Tensor, kernel, chunksize=None):
    """Maximum mean discrepancy with the reference's estimator (util.py:254-286).  For the two rotation
    kernels above, the three O(N^2) pair sums run in one fused HIP kernel each (no [N,N,3,3] intermediate, so
    `chunksize` is unnecessary and ignored); any other kernel callable goes through the reference's broadcast."""
    l_X, l_Y = len(X), len(Y)
    kind = _FUSED_KERNELS.get(kernel)
    if kind is not None:
        return (_b.kernel_sum(X, X, kind, 1.0 / (l_X ** 2)) + _b.kernel_sum(Y, Y, kind, 1.0 / (l_Y ** 2))
                - _b.kernel_sum(X, Y, kind, 2.0 / (l_X * l_Y)))
    X_sum = kernel(X.unsqueeze(0), X.unsqueeze(1)).sum(dim=(0, 1))
    Y_sum = kernel(Y.unsqueeze(0), Y.unsqueeze(1)).sum(dim=(0, 1))
    XY_sum = kernel(X.unsqueeze(0), Y.unsqueeze(1)).sum(dim=(0, 1))
    return (1 / (l_X ** 2)) * X_sum + (1 / (l_Y ** 2)) * Y_sum - (2 / (l_X * l_Y)) * XY_sum


def Ker_2samp_test(X, Y, kernel, alpha=0.05, max_ker=1, chunksize=None):
    """Kernel two-sample test (reference util.py:289-299): True = same distribution not rejected."""
    from math import log
    m, n = len(X), len(Y)
    assert m == n, "Requires equal amount of samples from X and Y"
    mmd = MMD(X, Y, kernel, chunksize=chunksize).item()
    return mmd < (2 * max_ker / m) ** 0.5 * (1 + (2 * log(1 / alpha)) ** 0.5)


def Ker_2samp_log_prob(X, Y, kernel, max_ker=1, chunksize=None):
    """log p-value bound of the test (reference util.py:301-312)"""
    m, n = len(X), len(Y)
    assert m == n, "Requires equal amount of samples from X and Y"
    mmd = MMD(X, Y, kernel, chunksize=chunksize).item()
    return -((((mmd / ((2 * max_ker / m) ** 0.5)) - 1) ** 2) / 2)
