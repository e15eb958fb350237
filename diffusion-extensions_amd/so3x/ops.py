"""Python-side registrations for the torch.ops.so3x operators that csrc/so3x_torch.cpp defines: fake-tensor (meta) kernels
-- what torch.compile, FakeTensorMode and torch.library.opcheck need to know the output shapes without running a kernel --
and the autograd formula of the score network's forward (its parameter gradient is the fused backward kernel; the rotation
inputs and timesteps carry no gradient on this path).  Imported by so3x.backend.ops() right after the
operator library is loaded."""
import torch
from torch.library import register_autograd, register_fake

_TRAP, _GUIDE_PITCH, _STASH_TILE = 999, 258, 17 * 1024


def _f32(like, shape):
    return like.new_empty(shape, dtype=torch.float32)


def _head(params):
    n = params.numel()
    return 3 if n == 17358 else 6


@register_fake("so3x::quat_to_rmat")
def _(q):
    return _f32(q, q.shape[:-1] + (3, 3))


@register_fake("so3x::log_rmat")
def _(R):
    return _f32(R, R.shape)


@register_fake("so3x::log_rmat_vec")
def _(R):
    return _f32(R, R.shape[:-2] + (3,))


@register_fake("so3x::exp_skewvec")
def _(v):
    return _f32(v, v.shape[:-1] + (3, 3))


@register_fake("so3x::orthogonalise")
def _(M):
    return _f32(M, M.shape)


@register_fake("so3x::so3_scale")
def _(R, k, k_stride):
    return _f32(R, R.shape)


@register_fake("so3x::aa_to_rmat")
def _(axis, ang):
    return _f32(axis, axis.shape[:-1] + (3, 3))


@register_fake("so3x::rmat_to_aa")
def _(R):
    return _f32(R, R.shape[:-2] + (3,)), _f32(R, R.shape[:-2] + (1,))


@register_fake("so3x::so3_lerp")
def _(a, a_stride, b, w, w_stride):
    return _f32(b, b.shape)


@register_fake("so3x::rmat_dist")
def _(a, b):
    return _f32(a, a.shape[:-2])


@register_fake("so3x::rmul")
def _(a, a_stride, b, b_stride, transpose_b):
    big = a if a.numel() >= b.numel() else b
    return _f32(big, big.shape)


@register_fake("so3x::igso3_eps_ft")
def _(omega, eps, eps_stride):
    return _f32(omega, omega.shape)


@register_fake("so3x::igso3_build_tables")
def _(eps):
    return _f32(eps, (eps.numel(), _TRAP))


@register_fake("so3x::igso3_build_guide")
def _(trap):
    return trap.new_empty((trap.numel() // _TRAP, _GUIDE_PITCH), dtype=torch.int16)


@register_fake("so3x::igso3_sample")
def _(trap, guide, row_idx, row_const, quirk_col0, axes, unif, seed, rng_offset, index_base, mean, n, want_angle, want_axis):
    return _f32(trap, (n, 3, 3)), _f32(trap, (n if want_angle else 0,)), _f32(trap, (n if want_axis else 0, 3))


@register_fake("so3x::igso3_logprob_score")
def _(R, eps, eps_stride, want_score, want_grad):
    lead = R.shape[:-2]
    return (_f32(R, lead + (1,)), _f32(R, lead + (3,) if want_score else (0, 3)), _f32(R, R.shape if want_grad else (0, 3, 3)))


@register_fake("so3x::mlp_fwd")
def _(params, x, t, t_stride, precision, t_table):
    return _f32(x, x.shape[:-2] + (_head(params),))


@register_fake("so3x::mlp_fwd_stash")
def _(params, x, t, t_stride, t_table):
    n = x.numel() // 9
    return _f32(x, x.shape[:-2] + (_head(params),)), x.new_empty(((n + 31) // 32 * _STASH_TILE,), dtype=torch.uint8)


@register_fake("so3x::mlp_bwd")
def _(params, x, t, t_stride, dout, precision, t_table, zstash):
    return _f32(x, (params.numel(),))


@register_fake("so3x::q_sample_target")
def _(sched, trap_q, guide_q, x0, t, quirk_col0, noise, axes, unif, seed, rng_offset, rng_offset_dev, index_base, want_x_t,
      want_target, want_noise):
    return (_f32(x0, x0.shape if want_x_t else (0, 3, 3)), _f32(x0, x0.shape[:-2] + (3,) if want_target else (0, 3)),
            _f32(x0, x0.shape if want_noise else (0, 3, 3)))


@register_fake("so3x::p_mean")
def _(sched, x, v, t, t_stride, t_const, want_x0hat):
    return _f32(x, x.shape if want_x0hat else (0, 3, 3)), _f32(x, x.shape)


@register_fake("so3x::p_sample_prepare")
def _(params, sched, trap_p, guide_p, precision):
    T = sched.shape[1]
    return params.new_empty((T * (4608 + 96 * 4 + 192 * 16) + (1 << 20),), dtype=torch.uint8)   # an upper bound is all a fake needs


@register_fake("so3x::p_sample_prepared")
def _(workspace, sched, trap_p, guide_p, x, t_start, t_dev, n_steps, axes, unif, seed, rng_offset, index_base, precision):
    return _f32(x, x.shape)


@register_fake("so3x::p_sample_prepared_out")
def _(workspace, sched, trap_p, guide_p, x, t_start, t_dev, n_steps, axes, unif, seed, rng_offset, index_base, precision, out):
    return None


@register_fake("so3x::resnet_p_sample_prepare")
def _(params, T, precision):
    return params.new_empty((4 * params.numel() + T * 1024 + (1 << 20),), dtype=torch.uint8)   # an upper bound is all a fake needs


@register_fake("so3x::resnet_p_sample_prepared")
def _(workspace, sched, trap_p, guide_p, x, t_start, t_dev, n_steps, axes, unif, seed, rng_offset, index_base, precision):
    return _f32(x, x.shape)


@register_fake("so3x::resnet_p_sample_prepared_out")
def _(workspace, sched, trap_p, guide_p, x, t_start, t_dev, n_steps, axes, unif, seed, rng_offset, index_base, precision, out):
    return None


@register_fake("so3x::p_sample_chain")
def _(params, sched, trap_p, guide_p, x, t_start, n_steps, axes, unif, seed, rng_offset, index_base, precision):
    return _f32(x, x.shape)


@register_fake("so3x::p_sample_chain_out")
def _(params, sched, trap_p, guide_p, x, t_start, n_steps, axes, unif, seed, rng_offset, index_base, precision, out):
    return None


@register_fake("so3x::train_fwd")
def _(params, sched, trap_q, guide_q, x0, t, quirk_col0, axes, unif, seed, rng_offset, rng_counter, index_base, want_out):
    n = x0.numel() // 9
    by = lambda k: x0.new_empty((k,), dtype=torch.uint8)  # noqa: E731
    return (_f32(x0, (1,)), _f32(x0, x0.shape), x0.new_empty((n,), dtype=torch.int64), _f32(x0, (n, 3)),
            by((n + 31) // 32 * _STASH_TILE), by(_train_workspace_bytes(n, sched.shape[1])), _f32(x0, (n if want_out else 0, 3)))


def _train_workspace_bytes(n, T):
    """so3x_train_workspace_bytes (a host-side size function of the C ABI); symbolic sizes get an upper bound"""
    try:
        import ctypes as C
        from .backend import lib
        return int(lib().so3x_train_workspace_bytes(C.c_int64(int(n)), C.c_int(int(T))))
    except Exception:  # noqa: BLE001
        return 32 << 20


@register_fake("so3x::train_bwd")
def _(x_t, t, dout, zstash, workspace, T, gscale, n_params):
    return _f32(x_t, (n_params,))


@register_fake("so3x::train_fused")
def _(params, sched, trap_q, guide_q, x0, t, quirk_col0, axes, unif, seed, rng_offset, rng_counter, index_base, loss, t_used, x_t, out, workspace):
    return None


@register_fake("so3x::train_noise")
def _(sched, trap_q, guide_q, x0, t, quirk_col0, axes, unif, seed, rng_offset, rng_counter, index_base, x_t, t_used, workspace):
    return None


@register_fake("so3x::train_net")
def _(params, T, x_t, t_used, dout, zstash, loss, out, rng_counter, workspace):
    return None


@register_fake("so3x::train_bwd_partial")
def _(x_t, t, dout, zstash, T, workspace):
    return None


@register_fake("so3x::train_bwd_reduce")
def _(n, T, gscale, grad, workspace):
    return None


@register_fake("so3x::train_bwd_reduce_adam")
def _(n, T, gscale, grad, workspace, params, exp_avg, exp_avg_sq, step, lr, beta1, beta2, eps, weight_decay, grad_scale):
    return None


@register_fake("so3x::adam_step")
def _(params, grad, exp_avg, exp_avg_sq, step, lr, beta1, beta2, eps, weight_decay, grad_scale):
    return None


# ---- autograd of the score network's forward: d loss / d params through so3x_mlp_bwd (the forward is recomputed inside the
#      backward kernels; the training path proper parks the pre-activations instead: so3x_train_fwd / mlp_fwd_stash)
def _mlp_setup(ctx, inputs, output):
    params, x, t, t_stride, precision, t_table = inputs
    ctx.save_for_backward(params, x, t)
    ctx.meta = (t_stride, precision, t_table)


def _mlp_backward(ctx, dout):
    params, x, t = ctx.saved_tensors
    t_stride, precision, t_table = ctx.meta
    dparams = torch.ops.so3x.mlp_bwd(params, x, t, t_stride, dout.contiguous().reshape(-1, dout.shape[-1]), precision, t_table, None)
    return dparams, None, None, None, None, None


register_autograd("so3x::mlp_fwd", _mlp_backward, setup_context=_mlp_setup)

# ---- widened rows
_RESNET_STASH = {0: 13 * 1024, 1: 13 * 512}  # bytes per sample: an upper bound is all a fake needs


@register_fake("so3x::rotate_cloud")
def _(rot, cloud, cloud_stride, P):
    return _f32(rot, rot.shape[:-2] + (P, 3))


@register_fake("so3x::planenet_prepare")
def _(params, dim, heads, layers, ffn, precision):
    return params.new_empty((3 * params.numel() + (1 << 20) if precision == 1 else 0,), dtype=torch.uint8)   # (an upper bound is all a fake needs)


@register_fake("so3x::planenet_fwd")
def _(params, x, t, dim, heads, layers, ffn, precision, want_stash, want_encoding, prepared, dropout_p, seed, rng_offset):
    B, P = x.shape[0], x.shape[1]
    # an upper bound is all a fake needs: per token and layer 7 dim + ffn + heads * P floats
    stash = (B * P * layers * (8 * dim + ffn + heads * P) * 4 + (1 << 20)) if want_stash else 0
    return _f32(x, (B, 3)), x.new_empty((stash,), dtype=torch.uint8), _f32(x, (B if want_encoding else 0, P, dim))


@register_fake("so3x::planenet_bwd")
def _(params, x, t, dout, stash, dim, heads, layers, ffn, precision, dropout_p, seed, rng_offset):
    return _f32(x, (params.numel(),))


@register_fake("so3x::protnet_fwd")
def _(params, rec_res, rec_pos, rec_ang, rec_off, lig_res, lig_pos, lig_ang, lig_off, t, max_len, dim, heads, t_depth, c_depth, precision, want_stash,
      want_pool, want_encoding, dropout_p, seed, rng_offset):
    B = t.numel()
    # an upper bound is all a fake needs: per padded token and layer 8 dim + 2048 + heads * max_len floats
    stash = (2 * B * max_len * (t_depth * (8 * dim + 2048 + heads * max_len) + 2 * c_depth * dim + 64) * 4 + (1 << 20)) if want_stash else 0
    return (_f32(rec_pos, (B, 6)), rec_pos.new_empty((stash,), dtype=torch.uint8), _f32(rec_pos, (B if want_pool else 0, 3 * dim + 6)),
            _f32(rec_pos, (2 * B if want_encoding else 0, max_len, dim)))


@register_fake("so3x::protnet_bwd")
def _(params, dout, stash, max_len, dim, heads, t_depth, c_depth, precision, dropout_p, seed, rng_offset):
    return _f32(dout, (params.numel(),))


@register_fake("so3x::resnet_fwd")
def _(params, x, t, t_stride, n_out, precision, t_table):
    return _f32(x, x.shape[:-2] + (n_out,))


@register_fake("so3x::resnet_fwd_stash")
def _(params, x, t, t_stride, n_out, precision, t_table):
    n = x.numel() // 9
    return _f32(x, x.shape[:-2] + (n_out,)), x.new_empty((n * _RESNET_STASH.get(precision, 13 * 1024),), dtype=torch.uint8)


@register_fake("so3x::resnet_bwd")
def _(params, x, t, t_stride, dout, n_out, precision, t_table, stash):
    return _f32(x, (params.numel(),))


@register_fake("so3x::resnet_p_sample_chain")
def _(params, sched, trap_p, guide_p, x, t_start, n_steps, axes, unif, seed, rng_offset, index_base, precision):
    return _f32(x, x.shape)


@register_fake("so3x::resnet_p_sample_chain_out")
def _(params, sched, trap_p, guide_p, x, t_start, n_steps, axes, unif, seed, rng_offset, index_base, precision, out):
    return None


@register_fake("so3x::se3_q_sample_target")
def _(sched, trap_q, guide_q, shift_scale, x0_rot, x0_shift, t, quirk_col0, axes, unif, znorm, seed, rng_offset, index_base, want_targets):
    n = x0_rot.numel() // 9
    k = n if want_targets else 0
    return _f32(x0_rot, x0_rot.shape), _f32(x0_rot, (n, 3)), _f32(x0_rot, (k, 3)), _f32(x0_rot, (k, 3))


@register_fake("so3x::se3_p_mean")
def _(sched, x_rot, x_shift, v_rot, v_shift, t):
    return _f32(x_rot, x_rot.shape), _f32(x_shift, x_shift.shape)


@register_fake("so3x::se3_p_noise")
def _(trap_row, sigma, shift_scale, mean_rot, mean_shift, axes, unif, znorm, seed, rng_offset, index_base, shared_rot):
    return _f32(mean_rot, mean_rot.shape), _f32(mean_shift, mean_shift.shape)


@register_fake("so3x::rigid_move")
def _(rot, shift, pos, frames):
    return _f32(pos, pos.shape), _f32(pos, frames.shape if frames is not None else (0, 3, 3))


@register_fake("so3x::rigid_move_ragged")
def _(rot, shift, pos, frames, off):
    return _f32(pos, pos.shape), _f32(pos, frames.shape if frames is not None else (0, 3, 3))


@register_fake("so3x::kernel_sum")
def _(X, Y, kind, scale):
    return _f32(X, (1,))


@register_fake("so3x::six2rmat")
def _(x6):
    return _f32(x6, x6.shape[:-1] + (3, 3))


@register_fake("so3x::six2rmat_bwd")
def _(x6, dR):
    return _f32(x6, x6.shape)


@register_fake("so3x::log_rmat_bwd")
def _(R, dlog):
    return _f32(R, R.shape)


@register_fake("so3x::rmat_dist_bwd")
def _(a, b, ddist):
    return _f32(a, a.shape), _f32(b, b.shape)


@register_fake("so3x::prevstep_loss")
def _(sched, x_recon, x_start, x_noisy, t, t_stride, want_dx, want_step):
    return (_f32(x_recon, (1,)), _f32(x_recon, x_recon.shape if want_dx else (0, 3, 3)), _f32(x_recon, x_recon.shape if want_step else (0, 3, 3)))


@register_fake("so3x::prevstep_loss6")
def _(sched, out6, x_start, x_noisy, t, t_stride):
    return _f32(out6, (1,)), _f32(out6, out6.shape)


@register_fake("so3x::mse_loss")
def _(a, b):
    return _f32(a, (1,))


@register_fake("so3x::mse_grad")
def _(a, b, gscale):
    return _f32(a, a.shape)


# ---- autograd of six2rmat (reference util.py:67-76 under torch autograd): the closed-form backward kernel
def _six_setup(ctx, inputs, output):
    ctx.save_for_backward(inputs[0])


def _six_backward(ctx, g):
    (x6,) = ctx.saved_tensors
    return torch.ops.so3x.six2rmat_bwd(x6, g.contiguous())


register_autograd("so3x::six2rmat", _six_backward, setup_context=_six_setup)
