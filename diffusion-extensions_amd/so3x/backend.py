"""Binding of libso3x.so (include/so3x.h) for torch tensors on an MI355X.

Every device entry point -- the SO(3) hot path (rotation algebra, IGSO(3), the score MLP, the diffusion
steps, the training step) and the widened rows (SE(3) layer, statistics, the 255-wide network, the rotation-matrix head)
-- is bound as a PyTorch-ROCm custom operator: libso3x_torch.so registers TORCH_LIBRARY(so3x, ...) over the C ABI
(csrc/so3x_torch.cpp) and the functions below call torch.ops.so3x.*.  ctypes is left for the four host-side table
builders (schedules, IGSO(3) knots, embedding frequencies: numpy in, numpy out) and the ABI / symbol check at load time.

PyTorch is plumbing here: it owns device memory and the HIP stream; every computation on the hot path happens in the
hand-written HIP kernels behind the C ABI.  There is NO CPU path and NO fallback: a missing library, a missing symbol, a
CPU tensor or a failed launch raises.
"""
import ctypes as C
import os
import threading

import numpy as np

import torch

_PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.path.join(_PKG, "libso3x.so")
TORCH_LIB_PATH = os.path.join(_PKG, "libso3x_torch.so")
HEADER_PATH = os.path.join(os.path.dirname(_PKG), "include", "so3x.h")

PREC_F32 = 0
PREC_BF16 = 1
PREC_F16 = 2  # so3x_p_sample_chain only: the bf16 path with IEEE half operand bits (a labelled extra leg, round 4)
N_PARAMS = 17358
N_PARAMS_ROTMAT = 17556  # out_type="rotmat": Linear(65, 6) head (reference so3_train.py:21-22)
SCHED_ROWS = 13
TRAP = 999
GUIDE_PITCH = 258

# every entry point declared in include/so3x.h (checked against the header by the tests)
SYMBOLS = (
    "so3x_abi_version", "so3x_error_string", "so3x_schedule_from_betas", "so3x_cosine_beta_schedule",
    "so3x_igso3_knots", "so3x_posemb_freqs", "so3x_quat_to_rmat", "so3x_log_rmat", "so3x_log_rmat_vec",
    "so3x_exp_skewvec", "so3x_orthogonalise", "so3x_so3_scale", "so3x_aa_to_rmat", "so3x_rmat_to_aa", "so3x_so3_lerp",
    "so3x_rmat_dist", "so3x_rmul", "so3x_igso3_eps_ft", "so3x_igso3_build_tables", "so3x_igso3_build_guide", "so3x_igso3_sample",
    "so3x_igso3_logprob_score", "so3x_mlp_workspace_bytes", "so3x_mlp_fwd", "so3x_mlp_bwd", "so3x_mlp_stash_bytes",
    "so3x_mlp_fwd_stash",
    "so3x_q_sample_target", "so3x_p_mean", "so3x_p_mean_t", "so3x_p_sample_workspace_bytes", "so3x_p_sample_chain",
    "so3x_se3_q_sample_target", "so3x_se3_p_mean", "so3x_se3_p_noise", "so3x_rigid_move", "so3x_rigid_move_ragged", "so3x_rotate_cloud",
    "so3x_kernel_sum_workspace_bytes", "so3x_kernel_sum", "so3x_mse_workspace_bytes", "so3x_mse_loss", "so3x_mse_grad",
    "so3x_resnet_workspace_bytes", "so3x_resnet_fwd", "so3x_resnet_p_sample_chain",
    "so3x_resnet_train_workspace_bytes", "so3x_resnet_bwd", "so3x_resnet_stash_bytes", "so3x_resnet_fwd_stash",
    "so3x_six2rmat", "so3x_six2rmat_bwd", "so3x_log_rmat_bwd", "so3x_rmat_dist_bwd", "so3x_prevstep_workspace_bytes",
    "so3x_prevstep_loss", "so3x_prevstep_loss6",
    "so3x_train_workspace_bytes", "so3x_train_fwd", "so3x_train_bwd", "so3x_adam_step",
    "so3x_p_sample_prepare", "so3x_p_sample_prepared", "so3x_resnet_p_sample_prepare", "so3x_resnet_p_sample_prepared",
    "so3x_planenet_weights_bytes", "so3x_planenet_prepare", "so3x_planenet_param_count", "so3x_planenet_workspace_bytes", "so3x_planenet_stash_bytes", "so3x_planenet_fwd", "so3x_planenet_bwd",
    "so3x_protnet_param_count", "so3x_protnet_workspace_bytes", "so3x_protnet_stash_bytes", "so3x_protnet_fwd", "so3x_protnet_bwd",
    "so3x_train_noise", "so3x_train_net", "so3x_train_bwd_partial", "so3x_train_bwd_reduce", "so3x_p_sample_clock_offset", "so3x_train_bwd_reduce_adam", "so3x_train_fused",
)


class So3xError(RuntimeError):
    pass


_lib = None
_lock = threading.RLock()  # ops() loads lib() under it


def lib():
    """Load libso3x.so (once).  Fails loudly: there is no alternative implementation."""
    global _lib
    if _lib is None:
        with _lock:
            if _lib is None:
                if not os.path.exists(LIB_PATH):
                    raise So3xError(
                        f"so3x: {LIB_PATH} not found -- build it with `make -C {os.path.join(_PKG, 'csrc')}` "
                        "(or __graft_entry__.build()); this backend has no CPU / PyTorch fallback")
                l = C.CDLL(LIB_PATH)
                missing = [s for s in SYMBOLS if not hasattr(l, s)]
                if missing:
                    raise So3xError(f"so3x: {LIB_PATH} lacks symbols {missing}")
                l.so3x_error_string.restype = C.c_char_p
                l.so3x_mlp_workspace_bytes.restype = C.c_size_t
                l.so3x_mlp_stash_bytes.restype = C.c_size_t
                l.so3x_p_sample_workspace_bytes.restype = C.c_size_t
                l.so3x_p_sample_clock_offset.restype = C.c_size_t
                l.so3x_kernel_sum_workspace_bytes.restype = C.c_size_t
                l.so3x_mse_workspace_bytes.restype = C.c_size_t
                l.so3x_resnet_workspace_bytes.restype = C.c_size_t
                l.so3x_resnet_train_workspace_bytes.restype = C.c_size_t
                l.so3x_resnet_stash_bytes.restype = C.c_size_t
                l.so3x_prevstep_workspace_bytes.restype = C.c_size_t
                l.so3x_train_workspace_bytes.restype = C.c_size_t
                l.so3x_planenet_workspace_bytes.restype = C.c_size_t
                l.so3x_planenet_stash_bytes.restype = C.c_size_t
                l.so3x_planenet_weights_bytes.restype = C.c_size_t
                l.so3x_planenet_param_count.restype = C.c_int64
                l.so3x_protnet_param_count.restype = C.c_int64
                l.so3x_protnet_workspace_bytes.restype = C.c_size_t
                l.so3x_protnet_stash_bytes.restype = C.c_size_t
                if l.so3x_abi_version() != 8:
                    raise So3xError("so3x: ABI version mismatch")
                _lib = l
    return _lib


_ops = None


def ops():
    """torch.ops.so3x after loading libso3x_torch.so (once).  Fails loudly: there is no alternative implementation."""
    global _ops
    if _ops is None:
        with _lock:
            if _ops is None:
                lib()  # ABI / symbol check of libso3x.so first
                if not os.path.exists(TORCH_LIB_PATH):
                    raise So3xError(f"so3x: {TORCH_LIB_PATH} not found -- build it with `make -C {os.path.join(_PKG, 'csrc')}` "
                                    "(or __graft_entry__.build()); this backend has no CPU / PyTorch fallback")
                torch.ops.load_library(TORCH_LIB_PATH)
                from . import ops as _register  # noqa: F401  (fake-tensor kernels of the operators)
                _ops = torch.ops.so3x
    return _ops


def _call(fn, *args):
    """a torch.ops.so3x call whose failures surface as So3xError, as the ctypes path's do"""
    try:
        return fn(*args)
    except RuntimeError as e:
        if "so3x:" in str(e):
            raise So3xError(str(e).split("\n")[0]) from None
        raise


def _s64(v):
    """an unsigned 64-bit seed / offset as the signed int a torch op schema carries"""
    v = int(v) & 0xFFFFFFFFFFFFFFFF
    return v - (1 << 64) if v >= (1 << 63) else v


def _check(rc, what):
    if rc != 0:
        raise So3xError(f"so3x: {what} failed: {lib().so3x_error_string(rc).decode()} (code {rc})")


def _dev(x, name, dtype=torch.float32):
    if not isinstance(x, torch.Tensor):
        raise TypeError(f"so3x: {name} must be a torch.Tensor")
    if not x.is_cuda:
        raise So3xError(f"so3x: {name} lives on {x.device}; the MI355X backend has no CPU path "
                        "(move the tensor/module to a 'cuda' device)")
    if x.dtype != dtype:
        x = x.to(dtype)
    return x if x.is_contiguous() else x.contiguous()


def _guide(g, trap, name="guide"):
    """optional uint16 search guide of `trap` (igso3_build_guide): same device, one 258-entry row per CDF row"""
    if g is None:
        return None
    g = _dev(g, name, torch.int16)
    if g.device != trap.device or g.numel() != (trap.numel() // TRAP) * GUIDE_PITCH:
        raise ValueError(f"so3x: {name} does not belong to this CDF table (build it with igso3_build_guide)")
    return g


def _out_like(out, x, name="out"):
    if out.dtype != torch.float32 or not out.is_cuda or out.device != x.device or not out.is_contiguous() or out.numel() != x.numel():
        raise ValueError(f"so3x: {name} must be a contiguous fp32 tensor of {x.numel()} elements on {x.device}")
    return out


# ----------------------------------------------------------------------------- host-side
def cosine_beta_schedule(T):
    import numpy as np
    out = np.empty(T, np.float64)
    _check(lib().so3x_cosine_beta_schedule(C.c_int(T), out.ctypes.data_as(C.c_void_p)), "cosine_beta_schedule")
    return out


def schedule_from_betas(betas):
    import numpy as np
    betas = np.ascontiguousarray(betas, np.float64)
    T = betas.shape[0]
    out = np.empty((SCHED_ROWS, T), np.float32)
    _check(lib().so3x_schedule_from_betas(betas.ctypes.data_as(C.c_void_p), C.c_int(T),
                                          out.ctypes.data_as(C.c_void_p)), "schedule_from_betas")
    return out


def igso3_knots():
    import numpy as np
    k = np.empty(1000, np.float32)
    w = np.empty(1000, np.float32)
    _check(lib().so3x_igso3_knots(k.ctypes.data_as(C.c_void_p), w.ctypes.data_as(C.c_void_p)), "igso3_knots")
    return k, w


def posemb_freqs(half_dim=28):
    import numpy as np
    out = np.empty(half_dim, np.float32)
    _check(lib().so3x_posemb_freqs(C.c_int(half_dim), out.ctypes.data_as(C.c_void_p)), "posemb_freqs")
    return out


# ----------------------------------------------------------------------------- rotations
def _rot_in(x, name):
    if x.shape[-2:] != (3, 3):
        raise ValueError(f"so3x: {name} must end in (3, 3), got {tuple(x.shape)}")
    return _dev(x, name)


def quat_to_rmat(q):
    return _call(ops().quat_to_rmat, _dev(q, "quaternions"))


def log_rmat(R):
    return _call(ops().log_rmat, _rot_in(R, "r_mat"))


def log_rmat_vec(R):
    return _call(ops().log_rmat_vec, _rot_in(R, "r_mat"))


def orthogonalise(M):
    """U round(S) V^T of each 3x3 matrix (reference util.py:95-107)"""
    return _call(ops().orthogonalise, _rot_in(M, "mat"))


def exp_skewvec(v):
    return _call(ops().exp_skewvec, _dev(v, "vec"))


def _per_sample(s, n, name, device):
    """scalar operand: numel 1 -> stride 0, numel n -> stride 1"""
    s = _dev(s if isinstance(s, torch.Tensor) else torch.tensor(s, device=device), name)
    if s.numel() == 1:
        return s.reshape(1), 0
    if s.numel() == n:
        return s.reshape(n), 1
    raise ValueError(f"so3x: {name} must have 1 or {n} elements, got {tuple(s.shape)}")


def so3_scale(R, scalars):
    R = _rot_in(R, "rmat")
    k, stride = _per_sample(scalars, R.numel() // 9, "scalars", R.device)
    return _call(ops().so3_scale, R, k, stride)


def aa_to_rmat(axis, ang):
    axis = _dev(axis, "rot_axis")
    n = axis.numel() // 3
    ang = _dev(ang, "ang")
    if ang.numel() != n:
        ang = ang.expand(axis.shape[:-1] + (1,)).contiguous()
    return _call(ops().aa_to_rmat, axis, ang)


def rmat_to_aa(R):
    return _call(ops().rmat_to_aa, _rot_in(R, "r_mat"))


def so3_lerp(a, b, w):
    b = _rot_in(b, "rot_b")
    if b.numel() == 9 and isinstance(w, torch.Tensor) and w.numel() > 1:
        # one end point, many weights (so3_lock_train.py:76-81: the arc between two fixed rotations): broadcast rot_b
        b = b.reshape(1, 3, 3).expand(w.numel(), 3, 3).contiguous()
    n = b.numel() // 9
    a = _rot_in(a, "rot_a")
    if a.numel() == 9:
        a_stride = 0
    elif a.numel() == b.numel():
        a_stride = 9
    else:
        raise ValueError("so3x: rot_a must be one (3,3) matrix or match rot_b")
    wt, w_stride = _per_sample(w, n, "weight", b.device)
    return _call(ops().so3_lerp, a, a_stride, b, wt, w_stride)


def rmat_dist(a, b):
    a = _rot_in(a, "input")
    b = _rot_in(b, "target")
    if a.shape != b.shape:
        a, b = torch.broadcast_tensors(a, b)
        a, b = a.contiguous(), b.contiguous()
    return _call(ops().rmat_dist, a, b)


def rmul(a, b, transpose_b=False):
    a = _rot_in(a, "a")
    b = _rot_in(b, "b")
    n = max(a.numel(), b.numel()) // 9
    if a.numel() != b.numel() and min(a.numel(), b.numel()) != 9:
        raise ValueError(f"so3x: rmul operands must match or one must be a single (3, 3) matrix, got {tuple(a.shape)} and {tuple(b.shape)}")
    sa = 0 if (a.numel() == 9 and n > 1) else 9
    sb = 0 if (b.numel() == 9 and n > 1) else 9
    return _call(ops().rmul, a, sa, b, sb, bool(transpose_b))


# ----------------------------------------------------------------------------- IGSO(3)
def igso3_eps_ft(omega, eps):
    omega = _dev(omega, "omega")
    e, stride = _per_sample(eps, omega.numel(), "eps", omega.device)
    return _call(ops().igso3_eps_ft, omega, e, stride)


def igso3_build_tables(eps):
    return _call(ops().igso3_build_tables, _dev(eps, "eps").reshape(-1))


def igso3_build_guide(trap):
    """uint16 [rows, 258] search guide of CDF rows (so3x_igso3_build_guide): pass it wherever rows are looked up per sample."""
    return _call(ops().igso3_build_guide, _dev(trap, "trap"))


def igso3_sample(trap, n, row_idx=None, row_const=0, quirk_col0=False, axes=None, unif=None, seed=0, rng_offset=0,
                 index_base=0, mean=None, want_angle=False, want_axis=False, guide=None):
    trap = _dev(trap, "trap")
    guide = _guide(guide, trap)
    ri = _dev(row_idx, "row_idx", torch.int64).reshape(-1) if row_idx is not None else None
    ax = _dev(axes, "axes").reshape(-1, 3) if axes is not None else None
    un = _dev(unif, "unif").reshape(-1) if unif is not None else None
    mn = _dev(mean, "mean").reshape(9) if mean is not None else None
    out, ang, axo = _call(ops().igso3_sample, trap, guide, ri, int(row_const), bool(quirk_col0), ax, un, _s64(seed), _s64(rng_offset),
                          int(index_base), mn, int(n), bool(want_angle), bool(want_axis))
    return out, (ang if want_angle else None), (axo if want_axis else None)


def igso3_logprob_score(R, eps, want_score=True, want_grad=False):
    R = _rot_in(R, "rotations")
    e, stride = _per_sample(eps, R.numel() // 9, "eps", R.device)
    logp, score, grad = _call(ops().igso3_logprob_score, R, e, stride, bool(want_score), bool(want_grad))
    return logp, (score if want_score else None), (grad if want_grad else None)


# ----------------------------------------------------------------------------- score MLP
def _t_arg(t, n):
    t = _dev(t, "t", torch.int64).reshape(-1)
    if t.numel() == 1:
        return t, 0
    if t.numel() == n:
        return t, 1
    raise ValueError(f"so3x: t must have 1 or {n} elements, got {t.numel()}")


def _head_width(numel, trunk, d, what):
    """3 (out_type "skewvec") or 6 ("rotmat") network outputs, read off the flat parameter count"""
    for k in (3, 6):
        if numel == trunk + k * (d + 1):
            return k
    raise ValueError(f"so3x: {what} params must hold {trunk + 3 * (d + 1)} (skewvec) or {trunk + 6 * (d + 1)} (rotmat) values")


def mlp_fwd(params, R, t, precision=PREC_F32, t_table=0):
    """RotPredict forward; returns the RAW network outputs [.., 3] or [.., 6] (six2rmat is a separate op)"""
    params = _dev(params, "params").reshape(-1)
    _head_width(params.numel(), N_PARAMS - 198, 65, "score-MLP")
    R = _rot_in(R, "x")
    tt, stride = _t_arg(t, R.numel() // 9)
    return _call(ops().mlp_fwd, params, R, tt, stride, int(precision), int(t_table))


def mlp_fwd_stash(params, R, t, t_table):
    """training forward (bf16 operands, bounded timesteps): (out, zstash) -- zstash goes to mlp_bwd(..., zstash=)"""
    params = _dev(params, "params").reshape(-1)
    _head_width(params.numel(), N_PARAMS - 198, 65, "score-MLP")
    R = _rot_in(R, "x")
    tt, stride = _t_arg(t, R.numel() // 9)
    return _call(ops().mlp_fwd_stash, params, R, tt, stride, int(t_table))


def mlp_bwd(params, R, t, dout, precision=PREC_F32, t_table=0, zstash=None):
    params = _dev(params, "params").reshape(-1)
    R = _rot_in(R, "x")
    tt, stride = _t_arg(t, R.numel() // 9)
    n_out = _head_width(params.numel(), N_PARAMS - 198, 65, "score-MLP")
    dout = _dev(dout, "dout").reshape(-1, n_out)
    return _call(ops().mlp_bwd, params, R, tt, stride, dout, int(precision), int(t_table), zstash)


# ----------------------------------------------------------------------------- diffusion
def q_sample_target(sched, trap_q, x0, t, quirk_col0=True, noise=None, axes=None, unif=None, seed=0, rng_offset=0,
                    index_base=0, want_x_t=True, want_target=True, want_noise=False, guide_q=None, rng_offset_dev=None):
    sched = _dev(sched, "sched")
    x0 = _rot_in(x0, "x_start")
    n = x0.numel() // 9
    tt = _dev(t, "t", torch.int64).reshape(-1)
    if tt.numel() != n:
        raise ValueError("so3x: t must have one entry per sample")
    tq = _dev(trap_q, "trap_q") if trap_q is not None else None
    guide_q = _guide(guide_q, tq, "guide_q") if tq is not None else None
    if rng_offset_dev is not None:
        rng_offset_dev = _dev(rng_offset_dev, "rng_offset_dev", torch.int64)
    nz = _rot_in(noise, "noise") if noise is not None else None
    ax = _dev(axes, "axes").reshape(-1, 3) if axes is not None else None
    un = _dev(unif, "unif").reshape(-1) if unif is not None else None
    x_t, tg, nzo = _call(ops().q_sample_target, sched, tq, guide_q, x0, tt, bool(quirk_col0), nz, ax, un, _s64(seed), _s64(rng_offset),
                         rng_offset_dev, int(index_base), bool(want_x_t), bool(want_target), bool(want_noise))
    return (x_t if want_x_t else None), (tg if want_target else None), (nzo if want_noise else None)


# ------------------------------------------------------------------ one training step (so3_train.py:73-76)
def train_fwd(params, sched, trap_q, x0, t, quirk_col0=True, axes=None, unif=None, seed=0, rng_offset=0, rng_counter=None,
              index_base=0, guide_q=None, want_out=False):
    """SO3Diffusion.p_losses for RotPredict(65, "skewvec") with bf16 operands in three launches (so3x_train_fwd): returns
    (loss [0-d], carry, out) where carry = (x_t, t, dout, zstash, workspace) is what train_bwd needs and out the network output
    when want_out.  t = None: the timesteps are drawn in the kernel from the samples' Philox blocks (and returned in the
    carry).  rng_counter: device int64 [1], read as an addend of rng_offset and incremented on the device."""
    params = _dev(params, "params").reshape(-1)
    if params.numel() != N_PARAMS:
        raise ValueError(f"so3x: the fused training step is built for the {N_PARAMS}-parameter skew-vector network")
    sched = _dev(sched, "sched")
    trap_q = _dev(trap_q, "trap_q")
    guide_q = _guide(guide_q, trap_q, "guide_q")
    x0 = _rot_in(x0, "x_start")
    n = x0.numel() // 9
    if n == 0:
        raise ValueError("so3x: empty batch")
    if t is not None:
        t = _dev(t, "t", torch.int64).reshape(-1)
        if t.numel() != n:
            raise ValueError("so3x: t must have one entry per sample")
    ax = _dev(axes, "axes").reshape(-1, 3) if axes is not None else None
    un = _dev(unif, "unif").reshape(-1) if unif is not None else None
    if (ax is None) != (un is None) or (ax is not None and (ax.shape[0] != n or un.numel() != n)):
        raise ValueError("so3x: axes [n, 3] and unif [n] go together")
    if rng_counter is not None:
        rng_counter = _dev(rng_counter, "rng_counter", torch.int64)
    loss, x_t, tt, dout, zstash, ws, out = _call(ops().train_fwd, params, sched, trap_q, guide_q, x0, t, bool(quirk_col0), ax, un, _s64(seed),
                                                 _s64(rng_offset), rng_counter, int(index_base), bool(want_out))
    return loss[0], (x_t, tt, dout, zstash, ws), (out if want_out else None)


def train_bwd(carry, n_params=N_PARAMS, T=None, gscale=None):
    """flat gradient [17358] of the loss train_fwd returned (so3x_train_bwd); gscale: 0-d / [1] device tensor or None"""
    x_t, tt, dout, zstash, ws = carry
    gs = _dev(gscale, "grad_output").reshape(1) if gscale is not None else None
    return _call(ops().train_bwd, x_t, tt, dout, zstash, ws, int(T), gs, int(n_params))


class TrainBuffers:
    """Caller-owned buffers of one training step (so3x_train_fused, or the stages so3x_train_noise / _net / _bwd_partial, and the
    reductions behind either): allocated once, reused by every step -- what a captured step (so3x.graphs.TrainStepGraph) runs on.
    staged=False: only what the one-kernel step needs (loss, workspace, flat gradient); x_t / t_used / out appear on demand."""

    def __init__(self, n, T, device, want_out=False, staged=True):
        l = lib()
        self.n, self.T = int(n), int(T)
        self.device = device
        f32 = dict(dtype=torch.float32, device=device)
        # (every kernel that fills these writes all of them: nothing to clear -- two fill launches less per eager step)
        self.loss = torch.zeros((1,), **f32) if staged else torch.empty((1,), **f32)
        self.workspace = torch.empty((int(l.so3x_train_workspace_bytes(C.c_int64(self.n), C.c_int(self.T))),), dtype=torch.uint8, device=device)
        self.grad = torch.zeros((N_PARAMS,), **f32) if staged else torch.empty((N_PARAMS,), **f32)
        self.x_t = self.t_used = self.dout = self.zstash = self.out = None
        if staged:
            self.x_t = torch.empty((self.n, 3, 3), **f32)
            self.t_used = torch.empty((self.n,), dtype=torch.int64, device=device)
            self.dout = torch.empty((self.n, 3), **f32)
            self.zstash = torch.empty((int(l.so3x_mlp_stash_bytes(C.c_int64(self.n))),), dtype=torch.uint8, device=device)
        if want_out:
            self.out = torch.empty((self.n, 3), **f32)

    # The eager training loop's buffers (one so3x_train_fused evaluation each): the workspace -- slabs, tickets, weight images,
    # a few MB -- is taken from a free list and handed back by the backward pass; the loss (returned to the caller) and the flat
    # gradient (it becomes the parameters' .grad) are fresh tensors every time.
    _free, _ws_bytes = {}, {}

    @classmethod
    def acquire(cls, n, T, device):
        key = (int(n), int(T), device)
        self = cls.__new__(cls)
        self.n, self.T, self.device, self._key = key[0], key[1], device, key
        self.loss = torch.empty((1,), dtype=torch.float32, device=device)
        self.grad = None
        free = cls._free.get(key)
        if torch.cuda.is_current_stream_capturing():
            free, self._key = None, None   # a captured step owns its workspace (graph-pool memory never enters the free list)
        if free:
            self.workspace = free.pop()
        else:
            nb = cls._ws_bytes.get(key[:2])
            if nb is None:
                nb = cls._ws_bytes[key[:2]] = int(lib().so3x_train_workspace_bytes(C.c_int64(self.n), C.c_int(self.T)))
            self.workspace = torch.empty((nb,), dtype=torch.uint8, device=device)
        self.x_t = self.t_used = self.dout = self.zstash = self.out = None
        return self

    @classmethod
    def release(cls, buf):
        key = getattr(buf, "_key", None)
        if key is not None and buf.workspace is not None and not torch.cuda.is_current_stream_capturing():
            free = cls._free.setdefault(key, [])
            if len(free) < 4:
                free.append(buf.workspace)
            buf.workspace = None


def train_noise(buf, sched, trap_q, x0, t=None, quirk_col0=True, axes=None, unif=None, seed=0, rng_offset=0, rng_counter=None,
                index_base=0, guide_q=None):
    """stage 1 of a training step: noise draw + q_sample + regression target for batch x0 into buf.x_t / buf.t_used / the target
    region of buf.workspace.  Depends on the data and the Philox counter only (not on the parameters)."""
    x0 = _rot_in(x0, "x_start")
    if x0.numel() // 9 != buf.n:
        raise ValueError("so3x: batch size differs from the buffers'")
    if t is not None:
        t = _dev(t, "t", torch.int64).reshape(-1)
    _call(ops().train_noise, _dev(sched, "sched"), _dev(trap_q, "trap_q"), _guide(guide_q, trap_q, "guide_q"), x0, t, bool(quirk_col0),
          _dev(axes, "axes").reshape(-1, 3) if axes is not None else None, _dev(unif, "unif").reshape(-1) if unif is not None else None,
          _s64(seed), _s64(rng_offset), rng_counter, int(index_base), buf.x_t, buf.t_used, buf.workspace)


def train_net(buf, params, rng_counter=None):
    """stage 2: weight images from the CURRENT params + network forward + stash + MSE and its gradient -> buf.loss, buf.dout;
    rng_counter (device int64 [1]) is advanced by one if given"""
    _call(ops().train_net, _dev(params, "params").reshape(-1), buf.T, buf.x_t, buf.t_used, buf.dout, buf.zstash, buf.loss, buf.out, rng_counter,
          buf.workspace)


def train_fused(buf, params, sched, trap_q, x0, t=None, quirk_col0=True, axes=None, unif=None, seed=0, rng_offset=0, rng_counter=None,
                index_base=0, guide_q=None, want_t=False, want_x_t=False, want_out=False):
    """noising + network forward + MSE + backward down to the partial dW slabs as ONE kernel (so3x_train_fused) into buf.loss and
    the slab region of buf.workspace; train_bwd_reduce / train_bwd_reduce_adam follows.  x_t, target, dout and the pre-activations
    never leave the chip; want_t / want_x_t / want_out (tests) fill buf.t_used / buf.x_t / buf.out."""
    x0 = _rot_in(x0, "x_start")
    if x0.numel() // 9 != buf.n:
        raise ValueError("so3x: batch size differs from the buffers'")
    if t is not None:
        t = _dev(t, "t", torch.int64).reshape(-1)
    if want_out and buf.out is None:
        buf.out = torch.empty((buf.n, 3), dtype=torch.float32, device=x0.device)
    if want_x_t and buf.x_t is None:
        buf.x_t = torch.empty((buf.n, 3, 3), dtype=torch.float32, device=x0.device)
    if want_t and buf.t_used is None:
        buf.t_used = torch.empty((buf.n,), dtype=torch.int64, device=x0.device)
    _call(ops().train_fused, _dev(params, "params").reshape(-1), _dev(sched, "sched"), _dev(trap_q, "trap_q"), _guide(guide_q, trap_q, "guide_q"),
          x0, t, bool(quirk_col0), _dev(axes, "axes").reshape(-1, 3) if axes is not None else None,
          _dev(unif, "unif").reshape(-1) if unif is not None else None, _s64(seed), _s64(rng_offset), rng_counter, int(index_base), buf.loss,
          buf.t_used if want_t else None, buf.x_t if want_x_t else None, buf.out if want_out else None, buf.workspace)


def train_bwd_partial(buf):
    """stage 3: the fused backward -> per-workgroup partial slabs in buf.workspace"""
    _call(ops().train_bwd_partial, buf.x_t, buf.t_used, buf.dout, buf.zstash, buf.T, buf.workspace)


def train_bwd_reduce(buf, gscale=None, grad=None):
    """stage 4: fixed-order sum of the slabs (x gscale) -> the flat gradient (buf.grad unless another tensor is given)"""
    g = buf.grad if grad is None else grad
    if g is None:
        g = buf.grad = torch.empty((N_PARAMS,), dtype=torch.float32, device=buf.device)
    _call(ops().train_bwd_reduce, buf.n, buf.T, _dev(gscale, "grad_output").reshape(1) if gscale is not None else None, g, buf.workspace)
    return g


def train_bwd_reduce_adam(buf, params, exp_avg, exp_avg_sq, step, lr, beta1, beta2, eps, weight_decay=0.0, grad_scale=1.0, gscale=None):
    """stage 4 and the optimizer in ONE launch (single-process training): buf.grad = the reduced gradient, then torch.optim.Adam's
    update of `params` with it -- bit-identical to train_bwd_reduce followed by adam_step"""
    from .flat import params_changed_out_of_band
    params_changed_out_of_band()   # the parameters are rewritten through a raw pointer: caches keyed on their values must see it
    _call(ops().train_bwd_reduce_adam, buf.n, buf.T, _dev(gscale, "grad_output").reshape(1) if gscale is not None else None, buf.grad,
          buf.workspace, params, exp_avg, exp_avg_sq, step, float(lr), float(beta1), float(beta2), float(eps), float(weight_decay),
          float(grad_scale))
    return buf.grad


def adam_step(params, grad, exp_avg, exp_avg_sq, step, lr, beta1, beta2, eps, weight_decay=0.0, grad_scale=1.0):
    """torch.optim.Adam.step() on flat fp32 buffers, in place (so3x_adam_step); step: device float32 [2] = [count, scratch]"""
    from .flat import params_changed_out_of_band
    params_changed_out_of_band()   # the parameters are rewritten through a raw pointer: caches keyed on their values must see it
    for name, x in (("params", params), ("grad", grad), ("exp_avg", exp_avg), ("exp_avg_sq", exp_avg_sq), ("step", step)):
        if not (isinstance(x, torch.Tensor) and x.is_cuda and x.dtype == torch.float32 and x.is_contiguous()):
            raise So3xError(f"so3x: adam_step needs contiguous fp32 device tensors ({name})")
    _call(ops().adam_step, params, grad, exp_avg, exp_avg_sq, step, float(lr), float(beta1), float(beta2), float(eps), float(weight_decay),
          float(grad_scale))


def p_mean(sched, x, v, t, want_x0hat=False):
    """x0hat (optional) and posterior mean for network output v (diffusion.py:291-313).  t: an int (one shared timestep), or
    an int64 tensor with one element (shared, read on the device: no host sync) or one per sample (the reference's
    extract(coef, t, ...))."""
    sched = _dev(sched, "sched")
    x = _rot_in(x, "x")
    v = _dev(v, "noise").reshape(-1, 3)
    n = x.numel() // 9
    if v.shape[0] != n:
        raise ValueError(f"so3x: the network output must be [{n}, 3], got {tuple(v.shape)}")
    if isinstance(t, torch.Tensor):
        tt, stride = _t_arg(t, n)
        x0h, mean = _call(ops().p_mean, sched, x, v, tt, stride, 0, bool(want_x0hat))
    else:
        x0h, mean = _call(ops().p_mean, sched, x, v, None, 0, int(t), bool(want_x0hat))
    return (x0h if want_x0hat else None), mean


def p_sample_prepare(params, sched, trap_p, precision=PREC_BF16, guide_p=None):
    """everything the reverse-chain kernel derives from the parameters and the tables, for all T timesteps: a byte tensor to hand
    to p_sample_prepared (valid until params / trap_p / guide_p change)"""
    params = _dev(params, "params").reshape(-1)
    if params.numel() != N_PARAMS:
        raise ValueError(f"so3x: params must hold {N_PARAMS} values")
    sched, trap_p = _dev(sched, "sched"), _dev(trap_p, "trap_p")
    return _call(ops().p_sample_prepare, params, sched, trap_p, _guide(guide_p, trap_p, "guide_p"), int(precision))


def resnet_p_sample_prepare(params, T, precision=PREC_BF16):
    """p_sample_prepare for the 255-wide residual network (so3x_resnet_p_sample_prepare): its weight image + input-row table"""
    return _call(ops().resnet_p_sample_prepare, _dev(params, "params").reshape(-1), int(T), int(precision))


def p_sample_prepared(ws, sched, trap_p, x, t_start, n_steps, t_dev=None, axes=None, unif=None, seed=0, rng_offset=0, index_base=0,
                      precision=PREC_BF16, out=None, guide_p=None, wide=False):
    """n_steps reverse steps from a prepared workspace: ONE launch.  t_dev (device int64, one element): the first timestep is read
    on the device instead of from t_start.  wide: the workspace is resnet_p_sample_prepare's (the 255-wide network)."""
    sched, trap_p = _dev(sched, "sched"), _dev(trap_p, "trap_p")
    x = _rot_in(x, "x")
    ax = _dev(axes, "axes").reshape(-1, 3) if axes is not None else None
    un = _dev(unif, "unif").reshape(-1) if unif is not None else None
    td = _dev(t_dev, "t", torch.int64).reshape(-1) if t_dev is not None else None
    args = (ws, sched, trap_p, _guide(guide_p, trap_p, "guide_p"), x, int(t_start), td, int(n_steps), ax, un, _s64(seed), _s64(rng_offset),
            int(index_base), int(precision))
    if out is None:
        return _call(ops().resnet_p_sample_prepared if wide else ops().p_sample_prepared, *args)
    _call(ops().resnet_p_sample_prepared_out if wide else ops().p_sample_prepared_out, *args, _out_like(out, x))
    return out


def p_sample_chain(params, sched, trap_p, x, t_start, n_steps, axes=None, unif=None, seed=0, rng_offset=0, index_base=0,
                   precision=PREC_BF16, out=None, guide_p=None):
    params = _dev(params, "params").reshape(-1)
    sched = _dev(sched, "sched")
    trap_p = _dev(trap_p, "trap_p")
    x = _rot_in(x, "x")
    ax = _dev(axes, "axes").reshape(-1, 3) if axes is not None else None
    un = _dev(unif, "unif").reshape(-1) if unif is not None else None
    guide_p = _guide(guide_p, trap_p, "guide_p")
    args = (params, sched, trap_p, guide_p, x, int(t_start), int(n_steps), ax, un, _s64(seed), _s64(rng_offset), int(index_base), int(precision))
    if out is None:
        return _call(ops().p_sample_chain, *args)
    _call(ops().p_sample_chain_out, *args, _out_like(out, x))
    return out


def rotate_cloud(rot, cloud):
    """cloud @ rot[..., 3, 3]^T -> [..., P, 3]  (PointCloudProj, reference models.py:75-91): cloud [P, 3] shared by every rotation,
    or [n, P, 3], one cloud per rotation (torch.matmul's batching, what aircraft_rotate.py feeds it)"""
    rot = _rot_in(rot, "x")
    cloud = _dev(cloud, "data")
    n = rot.numel() // 9
    if cloud.dim() == 2 or n == 0:
        cloud = cloud.reshape(-1, 3)
        P, stride = cloud.shape[0], 0
    else:
        if cloud.shape[:-2] != rot.shape[:-2]:
            raise ValueError(f"so3x: a batch of clouds {tuple(cloud.shape)} needs one rotation each, got {tuple(rot.shape)}")
        P = cloud.shape[-2]
        stride = 3 * P
    return _call(ops().rotate_cloud, rot, cloud, int(stride), int(P))


# ------------------------------------------------- wide residual score network (so3_lock_train.py:11-59)
N_PARAMS_RESNET = 6 * (255 * 255 + 255) + 3 * 255 + 3
N_PARAMS_RESNET_ROTMAT = 6 * (255 * 255 + 255) + 6 * 255 + 6
_RESNET_TRUNK = 6 * (255 * 255 + 255)


def resnet_fwd(params, R, t, t_table, precision=PREC_F32):
    params = _dev(params, "params").reshape(-1)
    n_out = _head_width(params.numel(), _RESNET_TRUNK, 255, "wide-net")
    R = _rot_in(R, "x")
    tt, stride = _t_arg(t, R.numel() // 9)
    return _call(ops().resnet_fwd, params, R, tt, stride, n_out, int(precision), int(t_table))


def resnet_fwd_stash(params, R, t, t_table, precision=PREC_BF16):
    """training forward: (out, stash) -- stash goes to resnet_bwd(..., stash=) instead of a second forward there"""
    params = _dev(params, "params").reshape(-1)
    n_out = _head_width(params.numel(), _RESNET_TRUNK, 255, "wide-net")
    R = _rot_in(R, "x")
    tt, stride = _t_arg(t, R.numel() // 9)
    return _call(ops().resnet_fwd_stash, params, R, tt, stride, n_out, int(precision), int(t_table))


def resnet_bwd(params, R, t, dout, t_table, precision=PREC_BF16, stash=None):
    """dL/dparams for dL/dout [n, 3 | 6]; stash = what resnet_fwd_stash returned for the same inputs, or None."""
    params = _dev(params, "params").reshape(-1)
    n_out = _head_width(params.numel(), _RESNET_TRUNK, 255, "wide-net")
    R = _rot_in(R, "x")
    tt, stride = _t_arg(t, R.numel() // 9)
    dout = _dev(dout, "dout").reshape(-1, n_out)
    return _call(ops().resnet_bwd, params, R, tt, stride, dout, n_out, int(precision), int(t_table), stash)


def resnet_p_sample_chain(params, sched, trap_p, x, t_start, n_steps, axes=None, unif=None, seed=0, rng_offset=0,
                          index_base=0, precision=PREC_BF16, out=None, guide_p=None):
    params = _dev(params, "params").reshape(-1)
    if params.numel() != N_PARAMS_RESNET:
        raise ValueError(f"so3x: params must hold {N_PARAMS_RESNET} values")
    sched = _dev(sched, "sched")
    trap_p = _dev(trap_p, "trap_p")
    x = _rot_in(x, "x")
    ax = _dev(axes, "axes").reshape(-1, 3) if axes is not None else None
    un = _dev(unif, "unif").reshape(-1) if unif is not None else None
    guide_p = _guide(guide_p, trap_p, "guide_p")
    args = (params, sched, trap_p, guide_p, x, int(t_start), int(n_steps), ax, un, _s64(seed), _s64(rng_offset), int(index_base), int(precision))
    if out is None:
        return _call(ops().resnet_p_sample_chain, *args)
    _call(ops().resnet_p_sample_chain_out, *args, _out_like(out, x))
    return out


# ----------------------------------------------------------------------------- PlaneNet (reference models.py:185-210)
def planenet_param_count(dim, heads, layers, ffn=2048):
    n = int(lib().so3x_planenet_param_count(C.c_int(dim), C.c_int(heads), C.c_int(layers), C.c_int(ffn)))
    if n < 0:
        raise ValueError(f"so3x: no PlaneNet with dim={dim}, heads={heads}, layers={layers}, ffn={ffn}")
    return n


def _planenet_in(params, x, t):
    params = _dev(params, "params").reshape(-1)
    x = _dev(x, "x")
    if x.dim() != 3 or x.shape[-1] != 3:
        raise ValueError("so3x: PlaneNet input must be [clouds, points, 3]")
    tt = _dev(t, "t", torch.int64).reshape(-1)
    if tt.numel() != x.shape[0]:
        raise ValueError("so3x: PlaneNet needs one timestep per cloud")
    return params, x, tt


def planenet_prepare(params, dim, heads, layers, ffn=2048, precision=PREC_F32):
    """the bf16 image of PlaneNet's weight matrices for planenet_fwd(prepared=...) (an empty tensor for the exact-fp32 form); valid
    until the parameters change"""
    return _call(ops().planenet_prepare, _dev(params, "params").reshape(-1), int(dim), int(heads), int(layers), int(ffn), int(precision))


def planenet_fwd(params, x, t, dim, heads, layers, ffn=2048, precision=PREC_F32, want_stash=False, want_encoding=False, prepared=None,
                 dropout_p=0.0, seed=0, rng_offset=0):
    """PlaneNet forward: (out [B, 3], stash for planenet_bwd or an empty tensor, encoder output [B, P, dim] or empty).
    dropout_p > 0: the training-mode forward (needs want_stash; masks are a function of (seed, rng_offset), see so3x.h)"""
    params, x, tt = _planenet_in(params, x, t)
    return _call(ops().planenet_fwd, params, x, tt, int(dim), int(heads), int(layers), int(ffn), int(precision), bool(want_stash), bool(want_encoding),
                 prepared, float(dropout_p), _s64(seed), _s64(rng_offset))


def planenet_bwd(params, x, t, dout, stash, dim, heads, layers, ffn=2048, precision=PREC_F32, dropout_p=0.0, seed=0, rng_offset=0):
    """d sum(out * dout) / d params (flat, state_dict order) from the stash planenet_fwd(want_stash=True) returned (and the SAME
    dropout_p, seed, rng_offset)"""
    params, x, tt = _planenet_in(params, x, t)
    dout = _dev(dout, "dout").reshape(-1, 3)
    return _call(ops().planenet_bwd, params, x, tt, dout, stash, int(dim), int(heads), int(layers), int(ffn), int(precision), float(dropout_p),
                 _s64(seed), _s64(rng_offset))


# ----------------------------------------------------------------------------- ProtNet (reference models.py:212-319)
def protnet_param_count(dim=64, heads=4, t_depth=4, c_depth=3):
    n = int(lib().so3x_protnet_param_count(C.c_int(dim), C.c_int(heads), C.c_int(t_depth), C.c_int(c_depth)))
    if n < 0:
        raise ValueError(f"so3x: no ProtNet with dim={dim}, heads={heads}, t_depth={t_depth}, c_depth={c_depth}")
    return n


class ProtBatch:
    """A batch of (receptor, ligand) ProtData pairs in the layout the kernels take (so3x.h): each field of each chain kind
    concatenated along the residue axis + int64 [B + 1] offsets; `max_len` = the longest chain.  Built once per batch
    (`ProtBatch.from_pairs`); `with_ligands` swaps in moved ligands (what ProtProjection produces every call) without touching the
    receptor side."""

    def __init__(self, rec, lig, rec_off, lig_off, max_len, lens):
        self.rec, self.lig, self.rec_off, self.lig_off, self.max_len, self.lens = rec, lig, rec_off, lig_off, int(max_len), lens

    @staticmethod
    def _cat(chains):
        res = torch.cat([_dev(c.residues, "residues") for c in chains]).contiguous()
        pos = torch.cat([_dev(c.positions, "positions") for c in chains]).contiguous()
        ang = torch.cat([_dev(c.angles, "angles").reshape(-1, 9) for c in chains]).contiguous()
        return res, pos, ang

    @classmethod
    def from_pairs(cls, pairs):
        pairs = list(pairs)
        lens = [(int(r.positions.shape[0]), int(l.positions.shape[0])) for r, l in pairs]
        dev = pairs[0][0].positions.device
        off = lambda k: torch.tensor([0] + list(np.cumsum([n[k] for n in lens])), dtype=torch.int64, device=dev)   # noqa: E731
        return cls(cls._cat([p[0] for p in pairs]), cls._cat([p[1] for p in pairs]), off(0), off(1), max(max(n) for n in lens), lens)

    def with_ligands(self, pos, ang):
        """the same batch with the ligands' positions [n_lig, 3] and frames [n_lig, 3, 3] replaced"""
        return ProtBatch(self.rec, (self.lig[0], _dev(pos, "positions").contiguous(), _dev(ang, "angles").reshape(-1, 9).contiguous()),
                         self.rec_off, self.lig_off, self.max_len, self.lens)

    def __len__(self):
        return len(self.lens)


def protnet_fwd(params, batch, t, dim=64, heads=4, t_depth=4, c_depth=3, precision=PREC_F32, want_stash=False, want_pool=False, want_encoding=False,
                dropout_p=0.0, seed=0, rng_offset=0):
    """ProtNet forward on a ProtBatch: (out [B, 6] = (rot_g, shift_g), stash for protnet_bwd or empty, the head's input [B, 3 dim + 6]
    or empty, rec_tf's output in the padded layout [2 B, max_len, dim] or empty).  dropout_p > 0: the training-mode forward (exact-fp32
    form, needs want_stash; masks are a function of (seed, rng_offset), see so3x.h)"""
    params = _dev(params, "params").reshape(-1)
    tt = _dev(t, "t", torch.int64).reshape(-1)
    if tt.numel() != len(batch):
        raise ValueError("so3x: ProtNet needs one timestep per complex")
    return _call(ops().protnet_fwd, params, *batch.rec, batch.rec_off, *batch.lig, batch.lig_off, tt, int(batch.max_len), int(dim), int(heads),
                 int(t_depth), int(c_depth), int(precision), bool(want_stash), bool(want_pool), bool(want_encoding), float(dropout_p), _s64(seed),
                 _s64(rng_offset))


def protnet_bwd(params, dout, stash, max_len, dim=64, heads=4, t_depth=4, c_depth=3, precision=PREC_F32, dropout_p=0.0, seed=0, rng_offset=0):
    """d sum(out * dout) / d params (flat, state_dict order; zero for lig_tf, which the reference never runs) from protnet_fwd's stash"""
    params = _dev(params, "params").reshape(-1)
    dout = _dev(dout, "dout").reshape(-1, 6).contiguous()
    return _call(ops().protnet_bwd, params, dout, stash, int(max_len), int(dim), int(heads), int(t_depth), int(c_depth), int(precision), float(dropout_p),
                 _s64(seed), _s64(rng_offset))


# ----------------------------------------------------------------------------- SE(3) layer
def se3_q_sample_target(sched, trap_q, shift_scale, x0_rot, x0_shift, t, quirk_col0=True, axes=None, unif=None, znorm=None,
                        seed=0, rng_offset=0, index_base=0, want_targets=True, guide_q=None):
    sched = _dev(sched, "sched")
    x0_rot = _rot_in(x0_rot, "x_start.rot")
    n = x0_rot.numel() // 9
    x0_shift = _dev(x0_shift, "x_start.shift").reshape(n, 3)
    trap_q = _dev(trap_q, "trap_q")
    guide_q = _guide(guide_q, trap_q, "guide_q")
    tt = _dev(t, "t", torch.int64).reshape(-1)
    ax = _dev(axes, "axes").reshape(-1, 3) if axes is not None else None
    un = _dev(unif, "unif").reshape(-1) if unif is not None else None
    zn = _dev(znorm, "znorm").reshape(-1, 3) if znorm is not None else None
    xt_rot, xt_shift, tg_rot, tg_shift = _call(ops().se3_q_sample_target, sched, trap_q, guide_q, float(shift_scale), x0_rot, x0_shift, tt,
                                               bool(quirk_col0), ax, un, zn, _s64(seed), _s64(rng_offset), int(index_base), bool(want_targets))
    return xt_rot, xt_shift, (tg_rot if want_targets else None), (tg_shift if want_targets else None)


def se3_p_mean(sched, x_rot, x_shift, v_rot, v_shift, t):
    sched = _dev(sched, "sched")
    x_rot = _rot_in(x_rot, "x.rot")
    n = x_rot.numel() // 9
    x_shift = _dev(x_shift, "x.shift").reshape(n, 3)
    v_rot = _dev(v_rot, "noise.rot_g").reshape(n, 3)
    v_shift = _dev(v_shift, "noise.shift_g").reshape(n, 3)
    return _call(ops().se3_p_mean, sched, x_rot, x_shift, v_rot, v_shift, int(t))


def se3_p_noise(trap_row, sigma, shift_scale, mean_rot, mean_shift, axes=None, unif=None, znorm=None, seed=0, rng_offset=0,
                index_base=0, shared_rot=True):
    mean_rot = _rot_in(mean_rot, "mean.rot")
    n = mean_rot.numel() // 9
    mean_shift = _dev(mean_shift, "mean.shift").reshape(n, 3)
    ax = _dev(axes, "axes").reshape(-1) if axes is not None else None
    un = _dev(unif, "unif").reshape(-1) if unif is not None else None
    zn = _dev(znorm, "znorm").reshape(-1, 3) if znorm is not None else None
    return _call(ops().se3_p_noise, _dev(trap_row, "trap_row"), float(sigma), float(shift_scale), mean_rot, mean_shift, ax, un, zn,
                 _s64(seed), _s64(rng_offset), int(index_base), bool(shared_rot))


def rigid_move(rot, shift, pos, frames=None):
    """rot [S,3,3], shift [S,3], pos [S,L,3], frames [S,L,3,3] or None."""
    rot = _rot_in(rot, "transf.rot")
    S = rot.numel() // 9
    shift = _dev(shift, "transf.shift").reshape(S, 3)
    pos = _dev(pos, "positions")
    fr = _dev(frames, "angles") if frames is not None else None
    out_pos, out_fr = _call(ops().rigid_move, rot, shift, pos, fr)
    return out_pos, (out_fr if fr is not None else None)


def rigid_move_ragged(rot, shift, pos, frames, off):
    """S structures of different lengths, concatenated: rot [S,3,3], shift [S,3], pos [n,3], frames [n,3,3] (or [n,9]) or None,
    off int64 [S + 1]; every structure moves about ITS OWN centroid (move_prot, prot_util.py:73-81)"""
    rot = _rot_in(rot, "transf.rot")
    S = rot.numel() // 9
    shift = _dev(shift, "transf.shift").reshape(S, 3)
    pos = _dev(pos, "positions")
    fr = _dev(frames, "angles") if frames is not None else None
    out_pos, out_fr = _call(ops().rigid_move_ragged, rot, shift, pos, fr, _dev(off, "offsets", torch.int64))
    return out_pos, (out_fr if fr is not None else None)


# ----------------------------------------------------------------------------- statistics
KERNEL_GAUSSIAN = 0
KERNEL_COSINE = 1


def kernel_sum(X, Y, kind=KERNEL_GAUSSIAN, scale=1.0):
    """scale * sum_ij k(X_i, Y_j) as a 0-d tensor (no host sync)."""
    X = _rot_in(X, "X").reshape(-1, 3, 3)
    Y = _rot_in(Y, "Y").reshape(-1, 3, 3)
    return _call(ops().kernel_sum, X, Y, int(kind), float(scale))[0]


# ------------------------------------------ rotation-matrix head + "prevstep" objective (8f row 3)
def six2rmat(x):
    """six2rmat (reference util.py:67-76); torch.ops.so3x.six2rmat carries its closed-form backward (so3x/ops.py)"""
    x = _dev(x, "x")
    if x.shape[-1] != 6:
        raise ValueError("so3x: six2rmat needs [..., 6]")
    return _call(ops().six2rmat, x)


def log_rmat_bwd(R, dlog):
    """autograd of log_rmat: dL/dlog [.., 3, 3] -> dL/dR"""
    return _call(ops().log_rmat_bwd, _rot_in(R, "r_mat"), _rot_in(dlog, "grad"))


def rmat_dist_bwd(a, b, ddist):
    """autograd of rmat_dist: dL/ddist [..] -> (dL/da, dL/db)"""
    return _call(ops().rmat_dist_bwd, _rot_in(a, "input"), _rot_in(b, "target"), _dev(ddist, "grad").reshape(-1))


class _LogRmat(torch.autograd.Function):
    @staticmethod
    def forward(ctx, R):
        ctx.save_for_backward(R)
        return log_rmat(R)

    @staticmethod
    def backward(ctx, g):
        return log_rmat_bwd(ctx.saved_tensors[0], g)


class _RmatDist(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        a = _rot_in(a, "input")
        b = _rot_in(b, "target")
        if a.shape != b.shape:
            raise ValueError("so3x: differentiable rmat_dist needs equal shapes")
        ctx.save_for_backward(a, b)
        return rmat_dist(a, b)

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        da, db = rmat_dist_bwd(a, b, g.contiguous())
        return da, db


def log_rmat_ad(R):
    """log_rmat that autograd can differentiate (reference util.py:164-192 under torch autograd)"""
    return _LogRmat.apply(R)


def rmat_dist_ad(a, b):
    """rmat_dist that autograd can differentiate (reference util.py:315-322 under torch autograd)"""
    return _RmatDist.apply(a, b)


class _PrevstepLoss(torch.autograd.Function):
    """mean_i rmat_dist(x_recon_i, x_noisy_i^T q_posterior_mean_i)^2 (reference diffusion.py:358-365), one fused kernel
    that also leaves d loss / d x_recon for the backward; x_start, x_noisy, t carry no gradient."""

    @staticmethod
    def forward(ctx, sched, x_recon, x_start, x_noisy, t):
        x_recon = _rot_in(x_recon, "x_recon")
        x_start = _rot_in(x_start, "x_start")
        x_noisy = _rot_in(x_noisy, "x_noisy")
        tt, stride = _t_arg(t, x_recon.numel() // 9)
        loss, dx, _ = _call(ops().prevstep_loss, _dev(sched, "sched"), x_recon, x_start, x_noisy, tt, stride, True, False)
        ctx.save_for_backward(dx)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        (dx,) = ctx.saved_tensors
        return None, dx * g, None, None, None


def prevstep_loss(sched, x_recon, x_start, x_noisy, t):
    return _PrevstepLoss.apply(sched, x_recon, x_start, x_noisy, t)


class _PrevstepLoss6(torch.autograd.Function):
    """_PrevstepLoss with x_recon = six2rmat(out6) applied inside: six2rmat, the loss and six2rmat's backward in one kernel"""

    @staticmethod
    def forward(ctx, sched, out6, x_start, x_noisy, t):
        out6 = _dev(out6, "out6").reshape(-1, 6)
        x_start = _rot_in(x_start, "x_start")
        x_noisy = _rot_in(x_noisy, "x_noisy")
        tt, stride = _t_arg(t, out6.shape[0])
        loss, d6 = _call(ops().prevstep_loss6, _dev(sched, "sched"), out6, x_start, x_noisy, tt, stride)
        ctx.save_for_backward(d6)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        (d6,) = ctx.saved_tensors
        return None, d6 * g, None, None, None


def prevstep_loss6(sched, out6, x_start, x_noisy, t):
    return _PrevstepLoss6.apply(sched, out6, x_start, x_noisy, t)


def prevstep_step(sched, x_start, x_noisy, t):
    """the rotation from x_noisy to the posterior mean of the previous step (reference diffusion.py:360-364)"""
    x_start = _rot_in(x_start, "x_start")
    x_noisy = _rot_in(x_noisy, "x_noisy")
    tt, stride = _t_arg(t, x_start.numel() // 9)
    return _call(ops().prevstep_loss, _dev(sched, "sched"), x_noisy, x_start, x_noisy, tt, stride, False, True)[2]


class _MSELoss(torch.autograd.Function):
    """mean((a - b)^2) with the gradient wrt `a` (the network output); `b` (the target) carries none."""

    @staticmethod
    def forward(ctx, a, b):
        a = _dev(a, "input")
        b = _dev(b, "target")
        if a.shape != b.shape:
            raise ValueError("so3x: mse_loss needs equal shapes")
        ctx.save_for_backward(a, b)
        return _call(ops().mse_loss, a, b)[0]

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        return _call(ops().mse_grad, a, b, _dev(g.reshape(1), "grad")), None


def mse_loss(a, b):
    """F.mse_loss(a, b) (reference diffusion.py:357) as a differentiable 0-d tensor."""
    return _MSELoss.apply(a, b)
