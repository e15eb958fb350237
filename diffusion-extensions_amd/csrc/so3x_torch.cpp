// so3x_torch.cpp -- PyTorch-ROCm custom operators over the C ABI of libso3x.so:  TORCH_LIBRARY(so3x, ...) schemas and
// their device implementations (dispatch key "CUDA", which is what PyTorch-ROCm calls the HIP device -- torch's key name,
// not a CUDA path).  Host-only C++ (g++): every op validates its tensors, allocates its outputs and workspaces from
// torch's caching allocator, takes the CURRENT HIP stream of the tensors' device and calls the extern "C" entry point
// that include/so3x.h declares.  Nothing is computed here.  Built into libso3x_torch.so by csrc/Makefile and loaded with
// torch.ops.load_library by so3x/backend.py; the Python classes that keep the reference's names (SO3Diffusion,
// IsotropicGaussianSO3, RotPredict, util.*) call torch.ops.so3x.* -- SURVEY.md 8b "Torch binding".
//
// Ops are functional (fresh outputs) unless the schema marks an argument (a!): the optimizer update and the device-
// resident counters.  Fake-tensor (meta) kernels and autograd formulas are registered from Python (so3x/ops.py).
#include <ATen/ATen.h>
#include <c10/core/DeviceGuard.h>
#include <c10/hip/HIPStream.h>
#include <torch/library.h>

#include "../../include/so3x.h"

namespace {

using at::Tensor;
using c10::optional;

void ok(int rc, const char* what) { TORCH_CHECK(rc == 0, "so3x: ", what, " failed: ", so3x_error_string(rc), " (code ", rc, ")"); }

const Tensor& dev(const Tensor& x, const char* name, at::ScalarType dt = at::kFloat) {
  TORCH_CHECK(x.is_cuda(), "so3x: ", name, " lives on ", x.device(), "; the MI355X backend has no CPU path");
  TORCH_CHECK(x.scalar_type() == dt, "so3x: ", name, " must be ", dt, ", got ", x.scalar_type());
  TORCH_CHECK(x.is_contiguous(), "so3x: ", name, " must be contiguous");
  return x;
}
so3x_stream_t strm(const Tensor& x) { return (so3x_stream_t)c10::hip::getCurrentHIPStream(x.device().index()).stream(); }
const float* F(const Tensor& x) { return x.const_data_ptr<float>(); }
float* Fm(Tensor& x) { return x.mutable_data_ptr<float>(); }
const float* Fo(const optional<Tensor>& x, const char* name) { return x.has_value() ? F(dev(*x, name)) : nullptr; }
const int64_t* I64(const Tensor& x) { return x.const_data_ptr<int64_t>(); }
const uint16_t* Guide(const optional<Tensor>& g) {
  return g.has_value() ? reinterpret_cast<const uint16_t*>(dev(*g, "guide", at::kShort).const_data_ptr<int16_t>()) : nullptr;
}
Tensor f32_like(const Tensor& like, at::IntArrayRef shape) { return at::empty(shape, like.options().dtype(at::kFloat)); }
Tensor bytes(const Tensor& like, size_t n) { return at::empty({(int64_t)n}, like.options().dtype(at::kByte)); }
std::vector<int64_t> with_tail(const Tensor& x, int drop, std::initializer_list<int64_t> tail) {
  std::vector<int64_t> s(x.sizes().begin(), x.sizes().end() - drop);
  s.insert(s.end(), tail.begin(), tail.end());
  return s;
}
#define GUARD(x) const c10::OptionalDeviceGuard device_guard(at::device_of(x))

// ------------------------------------------------------------------------------------------- rotation algebra
Tensor quat_to_rmat(const Tensor& q) {
  GUARD(q);
  Tensor out = f32_like(q, with_tail(q, 1, {3, 3}));
  ok(so3x_quat_to_rmat(strm(q), F(dev(q, "quaternions")), Fm(out), q.numel() / 4), "quat_to_rmat");
  return out;
}
#define UNARY_ROT(NAME, CFN, IN_TAIL, OUT_TAIL, INW)                                                  \
  Tensor NAME(const Tensor& x) {                                                                      \
    GUARD(x);                                                                                         \
    Tensor out = f32_like(x, with_tail(x, IN_TAIL, OUT_TAIL));                                        \
    ok(CFN(strm(x), F(dev(x, #NAME)), Fm(out), x.numel() / INW), #NAME);                              \
    return out;                                                                                       \
  }
UNARY_ROT(log_rmat, so3x_log_rmat, 2, (std::initializer_list<int64_t>{3, 3}), 9)
UNARY_ROT(log_rmat_vec, so3x_log_rmat_vec, 2, (std::initializer_list<int64_t>{3}), 9)
UNARY_ROT(exp_skewvec, so3x_exp_skewvec, 1, (std::initializer_list<int64_t>{3, 3}), 3)
UNARY_ROT(orthogonalise, so3x_orthogonalise, 2, (std::initializer_list<int64_t>{3, 3}), 9)

Tensor so3_scale(const Tensor& R, const Tensor& k, int64_t k_stride) {
  GUARD(R);
  Tensor out = at::empty_like(dev(R, "rmat"));
  ok(so3x_so3_scale(strm(R), F(R), F(dev(k, "scalars")), k_stride, Fm(out), R.numel() / 9), "so3_scale");
  return out;
}
Tensor aa_to_rmat(const Tensor& axis, const Tensor& ang) {
  GUARD(axis);
  Tensor out = f32_like(axis, with_tail(axis, 1, {3, 3}));
  ok(so3x_aa_to_rmat(strm(axis), F(dev(axis, "rot_axis")), F(dev(ang, "ang")), Fm(out), axis.numel() / 3), "aa_to_rmat");
  return out;
}
std::tuple<Tensor, Tensor> rmat_to_aa(const Tensor& R) {
  GUARD(R);
  Tensor axis = f32_like(R, with_tail(R, 2, {3})), ang = f32_like(R, with_tail(R, 2, {1}));
  ok(so3x_rmat_to_aa(strm(R), F(dev(R, "r_mat")), Fm(axis), Fm(ang), R.numel() / 9), "rmat_to_aa");
  return {axis, ang};
}
Tensor so3_lerp(const Tensor& a, int64_t a_stride, const Tensor& b, const Tensor& w, int64_t w_stride) {
  GUARD(b);
  Tensor out = at::empty_like(dev(b, "rot_b"));
  ok(so3x_so3_lerp(strm(b), F(dev(a, "rot_a")), a_stride, F(b), F(dev(w, "weight")), w_stride, Fm(out), b.numel() / 9), "so3_lerp");
  return out;
}
Tensor rmat_dist(const Tensor& a, const Tensor& b) {
  GUARD(a);
  Tensor out = f32_like(a, with_tail(a, 2, {}));
  ok(so3x_rmat_dist(strm(a), F(dev(a, "input")), F(dev(b, "target")), Fm(out), a.numel() / 9), "rmat_dist");
  return out;
}
Tensor rmul(const Tensor& a, int64_t a_stride, const Tensor& b, int64_t b_stride, bool transpose_b) {
  GUARD(a);
  const Tensor& big = a.numel() >= b.numel() ? a : b;
  Tensor out = at::empty_like(big);
  ok(so3x_rmul(strm(a), F(dev(a, "a")), a_stride, F(dev(b, "b")), b_stride, transpose_b ? 1 : 0, Fm(out), big.numel() / 9), "rmul");
  return out;
}

// ------------------------------------------------------------------------------------------------- IGSO(3)
Tensor igso3_eps_ft(const Tensor& omega, const Tensor& eps, int64_t eps_stride) {
  GUARD(omega);
  Tensor out = at::empty_like(dev(omega, "omega"));
  ok(so3x_igso3_eps_ft(strm(omega), F(omega), F(dev(eps, "eps")), eps_stride, Fm(out), omega.numel()), "igso3_eps_ft");
  return out;
}
Tensor igso3_build_tables(const Tensor& eps) {
  GUARD(eps);
  Tensor trap = f32_like(eps, {eps.numel(), SO3X_TRAP});
  ok(so3x_igso3_build_tables(strm(eps), F(dev(eps, "eps")), eps.numel(), Fm(trap)), "igso3_build_tables");
  return trap;
}
Tensor igso3_build_guide(const Tensor& trap) {
  GUARD(trap);
  const int64_t rows = trap.numel() / SO3X_TRAP;
  Tensor guide = at::empty({rows, SO3X_GUIDE_PITCH}, trap.options().dtype(at::kShort));
  ok(so3x_igso3_build_guide(strm(trap), F(dev(trap, "trap")), rows, reinterpret_cast<uint16_t*>(guide.mutable_data_ptr<int16_t>())),
     "igso3_build_guide");
  return guide;
}
std::tuple<Tensor, Tensor, Tensor> igso3_sample(const Tensor& trap, const optional<Tensor>& guide, const optional<Tensor>& row_idx,
                                                int64_t row_const, bool quirk_col0, const optional<Tensor>& axes,
                                                const optional<Tensor>& unif, int64_t seed, int64_t rng_offset, int64_t index_base,
                                                const optional<Tensor>& mean, int64_t n, bool want_angle, bool want_axis) {
  GUARD(trap);
  Tensor out = f32_like(trap, {n, 3, 3});
  Tensor ang = want_angle ? f32_like(trap, {n}) : Tensor();
  Tensor axo = want_axis ? f32_like(trap, {n, 3}) : Tensor();
  ok(so3x_igso3_sample(strm(trap), F(dev(trap, "trap")), Guide(guide), row_idx.has_value() ? I64(dev(*row_idx, "row_idx", at::kLong)) : nullptr,
                       row_const, quirk_col0 ? 1 : 0, Fo(axes, "axes"), Fo(unif, "unif"), (uint64_t)seed, (uint64_t)rng_offset, index_base,
                       Fo(mean, "mean"), Fm(out), want_angle ? Fm(ang) : nullptr, want_axis ? Fm(axo) : nullptr, n),
     "igso3_sample");
  return {out, ang.defined() ? ang : f32_like(trap, {0}), axo.defined() ? axo : f32_like(trap, {0, 3})};
}
std::tuple<Tensor, Tensor, Tensor> igso3_logprob_score(const Tensor& R, const Tensor& eps, int64_t eps_stride, bool want_score,
                                                       bool want_grad) {
  GUARD(R);
  Tensor logp = f32_like(R, with_tail(R, 2, {1}));
  Tensor score = f32_like(R, want_score ? with_tail(R, 2, {3}) : std::vector<int64_t>{0, 3});
  Tensor grad = want_grad ? at::empty_like(R) : f32_like(R, {0, 3, 3});
  ok(so3x_igso3_logprob_score(strm(R), F(dev(R, "rotations")), F(dev(eps, "eps")), eps_stride, Fm(logp), want_score ? Fm(score) : nullptr,
                              want_grad ? Fm(grad) : nullptr, R.numel() / 9),
     "igso3_logprob_score");
  return {logp, score, grad};
}

// ------------------------------------------------------------------------------------------------ score MLP
int64_t head_width(const Tensor& params) {
  const int64_t trunk = SO3X_MLP_PARAMS - 198;
  for (int64_t k : {3, 6})
    if (params.numel() == trunk + k * 66) return k;
  TORCH_CHECK(false, "so3x: score-MLP params must hold ", SO3X_MLP_PARAMS, " (skewvec) or ", SO3X_MLP_PARAMS_ROTMAT, " (rotmat) values");
}
Tensor mlp_fwd(const Tensor& params, const Tensor& R, const Tensor& t, int64_t t_stride, int64_t precision, int64_t t_table) {
  GUARD(R);
  const int64_t n = R.numel() / 9, n_out = head_width(params);
  Tensor out = f32_like(R, with_tail(R, 2, {n_out}));
  Tensor ws = bytes(R, so3x_mlp_workspace_bytes(0, (int)precision, (int)t_table));
  ok(so3x_mlp_fwd(strm(R), F(dev(params, "params")), F(dev(R, "x")), I64(dev(t, "t", at::kLong)), t_stride, Fm(out), n, (int)n_out,
                  (int)precision, (int)t_table, ws.mutable_data_ptr(), ws.numel()),
     "mlp_fwd");
  return out;
}
std::tuple<Tensor, Tensor> mlp_fwd_stash(const Tensor& params, const Tensor& R, const Tensor& t, int64_t t_stride, int64_t t_table) {
  GUARD(R);
  const int64_t n = R.numel() / 9, n_out = head_width(params);
  Tensor out = f32_like(R, with_tail(R, 2, {n_out}));
  Tensor zs = bytes(R, so3x_mlp_stash_bytes(n));
  Tensor ws = bytes(R, so3x_mlp_workspace_bytes(0, SO3X_PREC_BF16, (int)t_table));
  ok(so3x_mlp_fwd_stash(strm(R), F(dev(params, "params")), F(dev(R, "x")), I64(dev(t, "t", at::kLong)), t_stride, Fm(out),
                        zs.mutable_data_ptr(), n, (int)n_out, SO3X_PREC_BF16, (int)t_table, ws.mutable_data_ptr(), ws.numel()),
     "mlp_fwd_stash");
  return {out, zs};
}
Tensor mlp_bwd(const Tensor& params, const Tensor& R, const Tensor& t, int64_t t_stride, const Tensor& dout, int64_t precision,
               int64_t t_table, const optional<Tensor>& zstash) {
  GUARD(R);
  const int64_t n = R.numel() / 9, n_out = head_width(params);
  Tensor dparams = f32_like(R, {params.numel()});
  Tensor ws = bytes(R, so3x_mlp_workspace_bytes(n, (int)precision, (int)t_table));
  ok(so3x_mlp_bwd(strm(R), F(dev(params, "params")), F(dev(R, "x")), I64(dev(t, "t", at::kLong)), t_stride, F(dev(dout, "dout")),
                  Fm(dparams), n, (int)n_out, (int)precision, (int)t_table,
                  zstash.has_value() ? dev(*zstash, "zstash", at::kByte).const_data_ptr() : nullptr, ws.mutable_data_ptr(), ws.numel()),
     "mlp_bwd");
  return dparams;
}

// ------------------------------------------------------------------------------------------ diffusion steps
std::tuple<Tensor, Tensor, Tensor> q_sample_target(const Tensor& sched, const optional<Tensor>& trap_q, const optional<Tensor>& guide_q,
                                                   const Tensor& x0, const Tensor& t, bool quirk_col0, const optional<Tensor>& noise,
                                                   const optional<Tensor>& axes, const optional<Tensor>& unif, int64_t seed,
                                                   int64_t rng_offset, const optional<Tensor>& rng_offset_dev, int64_t index_base,
                                                   bool want_x_t, bool want_target, bool want_noise) {
  GUARD(x0);
  const int64_t n = x0.numel() / 9;
  const int T = (int)dev(sched, "sched").size(1);
  Tensor x_t = want_x_t ? at::empty_like(dev(x0, "x_start")) : f32_like(x0, {0, 3, 3});
  Tensor tg = f32_like(x0, want_target ? with_tail(x0, 2, {3}) : std::vector<int64_t>{0, 3});
  Tensor nz = want_noise ? at::empty_like(x0) : f32_like(x0, {0, 3, 3});
  ok(so3x_q_sample_target(strm(x0), F(sched), T, Fo(trap_q, "trap_q"), Guide(guide_q), F(x0), I64(dev(t, "t", at::kLong)),
                          quirk_col0 ? 1 : 0, Fo(noise, "noise"), Fo(axes, "axes"), Fo(unif, "unif"), (uint64_t)seed, (uint64_t)rng_offset,
                          rng_offset_dev.has_value() ? I64(dev(*rng_offset_dev, "rng_offset_dev", at::kLong)) : nullptr, index_base,
                          want_x_t ? Fm(x_t) : nullptr, want_target ? Fm(tg) : nullptr, want_noise ? Fm(nz) : nullptr, n),
     "q_sample_target");
  return {x_t, tg, nz};
}
std::tuple<Tensor, Tensor> p_mean(const Tensor& sched, const Tensor& x, const Tensor& v, const optional<Tensor>& t, int64_t t_stride,
                                  int64_t t_const, bool want_x0hat) {
  GUARD(x);
  const int T = (int)dev(sched, "sched").size(1);
  Tensor x0h = want_x0hat ? at::empty_like(dev(x, "x")) : f32_like(x, {0, 3, 3});
  Tensor mean = at::empty_like(x);
  if (t.has_value())
    ok(so3x_p_mean_t(strm(x), F(sched), T, F(x), F(dev(v, "noise")), I64(dev(*t, "t", at::kLong)), t_stride, want_x0hat ? Fm(x0h) : nullptr,
                     Fm(mean), x.numel() / 9),
       "p_mean");
  else
    ok(so3x_p_mean(strm(x), F(sched), T, F(x), F(dev(v, "noise")), (int)t_const, want_x0hat ? Fm(x0h) : nullptr, Fm(mean), x.numel() / 9),
       "p_mean");
  return {x0h, mean};
}
void chain_into(const Tensor& params, const Tensor& sched, const Tensor& trap_p, const optional<Tensor>& guide_p, const Tensor& x, Tensor& out,
                int64_t t_start, int64_t n_steps, const optional<Tensor>& axes, const optional<Tensor>& unif, int64_t seed, int64_t rng_offset,
                int64_t index_base, int64_t precision) {
  const int T = (int)dev(sched, "sched").size(1);
  TORCH_CHECK(out.numel() == x.numel() && out.device() == x.device(), "so3x: out must match x");
  Tensor ws = bytes(x, so3x_p_sample_workspace_bytes(T, (int)precision));
  ok(so3x_p_sample_chain(strm(x), F(dev(params, "params")), F(sched), T, F(dev(trap_p, "trap_p")), Guide(guide_p), F(dev(x, "x")),
                         Fm(const_cast<Tensor&>(dev(out, "out"))), (int)t_start, (int)n_steps, Fo(axes, "axes"), Fo(unif, "unif"), (uint64_t)seed,
                         (uint64_t)rng_offset, index_base, x.numel() / 9, (int)precision, ws.mutable_data_ptr(), ws.numel()),
     "p_sample_chain");
}
Tensor p_sample_chain(const Tensor& params, const Tensor& sched, const Tensor& trap_p, const optional<Tensor>& guide_p, const Tensor& x,
                      int64_t t_start, int64_t n_steps, const optional<Tensor>& axes, const optional<Tensor>& unif, int64_t seed,
                      int64_t rng_offset, int64_t index_base, int64_t precision) {
  GUARD(x);
  Tensor out = at::empty_like(x);
  chain_into(params, sched, trap_p, guide_p, x, out, t_start, n_steps, axes, unif, seed, rng_offset, index_base, precision);
  return out;
}
// the same into a caller-owned tensor, which may be x itself (the steps are applied in place)
void p_sample_chain_out(const Tensor& params, const Tensor& sched, const Tensor& trap_p, const optional<Tensor>& guide_p, const Tensor& x,
                        int64_t t_start, int64_t n_steps, const optional<Tensor>& axes, const optional<Tensor>& unif, int64_t seed,
                        int64_t rng_offset, int64_t index_base, int64_t precision, Tensor& out) {
  GUARD(x);
  chain_into(params, sched, trap_p, guide_p, x, out, t_start, n_steps, axes, unif, seed, rng_offset, index_base, precision);
}

// prepared-state sampling (so3x_p_sample_prepare / so3x_p_sample_prepared): the preparation as a tensor the caller keeps
Tensor p_sample_prepare(const Tensor& params, const Tensor& sched, const Tensor& trap_p, const optional<Tensor>& guide_p, int64_t precision) {
  GUARD(params);
  const int T = (int)dev(sched, "sched").size(1);
  Tensor ws = bytes(params, so3x_p_sample_workspace_bytes(T, (int)precision));
  ok(so3x_p_sample_prepare(strm(params), F(dev(params, "params")), T, F(dev(trap_p, "trap_p")), Guide(guide_p), (int)precision,
                           ws.mutable_data_ptr(), ws.numel()),
     "p_sample_prepare");
  return ws;
}
void p_sample_prepared_out(const Tensor& ws, const Tensor& sched, const Tensor& trap_p, const optional<Tensor>& guide_p, const Tensor& x, int64_t t_start,
                           const optional<Tensor>& t_dev, int64_t n_steps, const optional<Tensor>& axes, const optional<Tensor>& unif, int64_t seed,
                           int64_t rng_offset, int64_t index_base, int64_t precision, Tensor& out) {
  GUARD(x);
  const int T = (int)dev(sched, "sched").size(1);
  TORCH_CHECK(out.numel() == x.numel() && out.device() == x.device(), "so3x: out must match x");
  ok(so3x_p_sample_prepared(strm(x), F(sched), T, F(dev(trap_p, "trap_p")), Guide(guide_p), F(dev(x, "x")), Fm(const_cast<Tensor&>(dev(out, "out"))),
                            (int)t_start, t_dev.has_value() ? I64(dev(*t_dev, "t", at::kLong)) : nullptr, (int)n_steps, Fo(axes, "axes"), Fo(unif, "unif"),
                            (uint64_t)seed, (uint64_t)rng_offset, index_base, x.numel() / 9, (int)precision,
                            const_cast<Tensor&>(dev(ws, "workspace", at::kByte)).mutable_data_ptr(), ws.numel()),
     "p_sample_prepared");
}
Tensor p_sample_prepared(const Tensor& ws, const Tensor& sched, const Tensor& trap_p, const optional<Tensor>& guide_p, const Tensor& x, int64_t t_start,
                         const optional<Tensor>& t_dev, int64_t n_steps, const optional<Tensor>& axes, const optional<Tensor>& unif, int64_t seed,
                         int64_t rng_offset, int64_t index_base, int64_t precision) {
  GUARD(x);
  Tensor out = at::empty_like(x);
  p_sample_prepared_out(ws, sched, trap_p, guide_p, x, t_start, t_dev, n_steps, axes, unif, seed, rng_offset, index_base, precision, out);
  return out;
}

// the same for the 255-wide residual network (so3x_resnet_p_sample_prepare / _prepared)
Tensor resnet_p_sample_prepare(const Tensor& params, int64_t T, int64_t precision) {
  GUARD(params);
  Tensor ws = bytes(params, so3x_resnet_workspace_bytes((int)precision, (int)T));
  ok(so3x_resnet_p_sample_prepare(strm(params), F(dev(params, "params")), (int)T, (int)precision, ws.mutable_data_ptr(), ws.numel()), "resnet_p_sample_prepare");
  return ws;
}
void resnet_p_sample_prepared_out(const Tensor& ws, const Tensor& sched, const Tensor& trap_p, const optional<Tensor>& guide_p, const Tensor& x,
                                  int64_t t_start, const optional<Tensor>& t_dev, int64_t n_steps, const optional<Tensor>& axes,
                                  const optional<Tensor>& unif, int64_t seed, int64_t rng_offset, int64_t index_base, int64_t precision, Tensor& out) {
  GUARD(x);
  const int T = (int)dev(sched, "sched").size(1);
  TORCH_CHECK(out.numel() == x.numel() && out.device() == x.device(), "so3x: out must match x");
  ok(so3x_resnet_p_sample_prepared(strm(x), F(sched), T, F(dev(trap_p, "trap_p")), Guide(guide_p), F(dev(x, "x")), Fm(const_cast<Tensor&>(dev(out, "out"))),
                                   (int)t_start, t_dev.has_value() ? I64(dev(*t_dev, "t", at::kLong)) : nullptr, (int)n_steps, Fo(axes, "axes"),
                                   Fo(unif, "unif"), (uint64_t)seed, (uint64_t)rng_offset, index_base, x.numel() / 9, (int)precision,
                                   dev(ws, "workspace", at::kByte).const_data_ptr(), ws.numel()),
     "resnet_p_sample_prepared");
}
Tensor resnet_p_sample_prepared(const Tensor& ws, const Tensor& sched, const Tensor& trap_p, const optional<Tensor>& guide_p, const Tensor& x,
                                int64_t t_start, const optional<Tensor>& t_dev, int64_t n_steps, const optional<Tensor>& axes,
                                const optional<Tensor>& unif, int64_t seed, int64_t rng_offset, int64_t index_base, int64_t precision) {
  GUARD(x);
  Tensor out = at::empty_like(x);
  resnet_p_sample_prepared_out(ws, sched, trap_p, guide_p, x, t_start, t_dev, n_steps, axes, unif, seed, rng_offset, index_base, precision, out);
  return out;
}

// ---------------------------------------------------------------------------------------- one training step
// -> (loss[1], x_t, t, dout, zstash, workspace, out): everything so3x_train_bwd needs travels as tensors
std::tuple<Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor> train_fwd(
    const Tensor& params, const Tensor& sched, const Tensor& trap_q, const optional<Tensor>& guide_q, const Tensor& x0,
    const optional<Tensor>& t, bool quirk_col0, const optional<Tensor>& axes, const optional<Tensor>& unif, int64_t seed, int64_t rng_offset,
    optional<Tensor> rng_counter, int64_t index_base, bool want_out) {
  GUARD(x0);
  const int64_t n = x0.numel() / 9;
  const int T = (int)dev(sched, "sched").size(1);
  TORCH_CHECK(params.numel() == SO3X_MLP_PARAMS, "so3x: the fused training step is built for the ", SO3X_MLP_PARAMS, "-parameter skew-vector network");
  TORCH_CHECK(n > 0, "so3x: empty batch");
  Tensor tt = at::empty({n}, x0.options().dtype(at::kLong));  // the timesteps the step ran with: always a fresh tensor, never the caller's t
  Tensor x_t = at::empty_like(dev(x0, "x_start"));
  Tensor dout = f32_like(x0, {n, 3}), loss = f32_like(x0, {1});
  Tensor out = want_out ? f32_like(x0, {n, 3}) : f32_like(x0, {0, 3});
  Tensor zs = bytes(x0, so3x_mlp_stash_bytes(n)), ws = bytes(x0, so3x_train_workspace_bytes(n, T));
  ok(so3x_train_fwd(strm(x0), F(dev(params, "params")), F(sched), T, F(dev(trap_q, "trap_q")), Guide(guide_q), F(x0),
                    t.has_value() ? I64(dev(*t, "t", at::kLong)) : nullptr, tt.mutable_data_ptr<int64_t>(), quirk_col0 ? 1 : 0,
                    Fo(axes, "axes"), Fo(unif, "unif"), (uint64_t)seed, (uint64_t)rng_offset,
                    rng_counter.has_value() ? dev(*rng_counter, "rng_counter", at::kLong).mutable_data_ptr<int64_t>() : nullptr, index_base, n,
                    Fm(x_t), Fm(dout), zs.mutable_data_ptr(), Fm(loss), want_out ? Fm(out) : nullptr, ws.mutable_data_ptr(), ws.numel()),
     "train_fwd");
  return {loss, x_t, tt, dout, zs, ws, out};
}
Tensor train_bwd(const Tensor& x_t, const Tensor& t, const Tensor& dout, const Tensor& zstash, Tensor& workspace, int64_t T,
                 const optional<Tensor>& gscale, int64_t n_params) {
  GUARD(x_t);
  Tensor grad = f32_like(x_t, {n_params});
  Tensor& ws = workspace;  // the slabs region is written
  ok(so3x_train_bwd(strm(x_t), F(dev(x_t, "x_t")), I64(dev(t, "t", at::kLong)), F(dev(dout, "dout")), dev(zstash, "zstash", at::kByte).const_data_ptr(),
                    x_t.numel() / 9, (int)T, Fo(gscale, "grad_output"), Fm(grad), ws.mutable_data_ptr(), ws.numel()),
     "train_bwd");
  return grad;
}
// noising + forward + loss + backward down to the partial dW slabs as ONE kernel, on caller-owned buffers (train_bwd_reduce /
// train_bwd_reduce_adam with the same n, T, workspace follows)
void train_fused(const Tensor& params, const Tensor& sched, const Tensor& trap_q, const optional<Tensor>& guide_q, const Tensor& x0,
                 const optional<Tensor>& t, bool quirk_col0, const optional<Tensor>& axes, const optional<Tensor>& unif, int64_t seed,
                 int64_t rng_offset, optional<Tensor> rng_counter, int64_t index_base, Tensor& loss, optional<Tensor> t_used, optional<Tensor> x_t,
                 optional<Tensor> out, Tensor& workspace) {
  GUARD(x0);
  const int64_t n = x0.numel() / 9;
  const int T = (int)dev(sched, "sched").size(1);
  TORCH_CHECK(params.numel() == SO3X_MLP_PARAMS, "so3x: the fused training step is built for the ", SO3X_MLP_PARAMS, "-parameter skew-vector network");
  TORCH_CHECK(n > 0, "so3x: empty batch");
  TORCH_CHECK(loss.numel() >= 1 && (!t_used.has_value() || t_used->numel() == n) && (!x_t.has_value() || x_t->numel() == n * 9) &&
              (!out.has_value() || out->numel() == n * 3) && (!t.has_value() || t->numel() == n), "so3x: train_fused buffer sizes differ");
  dev(loss, "loss"); dev(workspace, "workspace", at::kByte);
  ok(so3x_train_fused(strm(x0), F(dev(params, "params")), F(sched), T, F(dev(trap_q, "trap_q")), Guide(guide_q), F(dev(x0, "x_start")),
                      t.has_value() ? I64(dev(*t, "t", at::kLong)) : nullptr,
                      t_used.has_value() ? dev(*t_used, "t_used", at::kLong).mutable_data_ptr<int64_t>() : nullptr, quirk_col0 ? 1 : 0,
                      Fo(axes, "axes"), Fo(unif, "unif"), (uint64_t)seed, (uint64_t)rng_offset,
                      rng_counter.has_value() ? dev(*rng_counter, "rng_counter", at::kLong).mutable_data_ptr<int64_t>() : nullptr, index_base, n,
                      Fm(loss), x_t.has_value() ? dev(*x_t, "x_t").mutable_data_ptr<float>() : nullptr,
                      out.has_value() ? dev(*out, "out").mutable_data_ptr<float>() : nullptr, workspace.mutable_data_ptr(), workspace.numel()),
     "train_fused");
}
// the step in stages, on caller-owned buffers (so3x.graphs.TrainStepGraph pipelines them across two streams)
void train_noise(const Tensor& sched, const Tensor& trap_q, const optional<Tensor>& guide_q, const Tensor& x0, const optional<Tensor>& t,
                 bool quirk_col0, const optional<Tensor>& axes, const optional<Tensor>& unif, int64_t seed, int64_t rng_offset,
                 const optional<Tensor>& rng_counter, int64_t index_base, Tensor& x_t, Tensor& t_used, Tensor& workspace) {
  GUARD(x0);
  const int64_t n = x0.numel() / 9;
  const int T = (int)dev(sched, "sched").size(1);
  TORCH_CHECK(n > 0 && x_t.numel() == n * 9 && t_used.numel() == n, "so3x: train_noise buffer sizes differ");
  dev(x_t, "x_t"); dev(workspace, "workspace", at::kByte);
  ok(so3x_train_noise(strm(x0), F(sched), T, F(dev(trap_q, "trap_q")), Guide(guide_q), F(dev(x0, "x_start")),
                      t.has_value() ? I64(dev(*t, "t", at::kLong)) : nullptr, dev(t_used, "t_used", at::kLong).mutable_data_ptr<int64_t>(),
                      quirk_col0 ? 1 : 0, Fo(axes, "axes"), Fo(unif, "unif"), (uint64_t)seed, (uint64_t)rng_offset,
                      rng_counter.has_value() ? I64(dev(*rng_counter, "rng_counter", at::kLong)) : nullptr, index_base, n, Fm(x_t),
                      workspace.mutable_data_ptr(), workspace.numel()),
     "train_noise");
}
void train_net(const Tensor& params, int64_t T, const Tensor& x_t, const Tensor& t_used, Tensor& dout, Tensor& zstash, Tensor& loss,
               optional<Tensor> out, optional<Tensor> rng_counter, Tensor& workspace) {
  GUARD(x_t);
  const int64_t n = x_t.numel() / 9;
  TORCH_CHECK(params.numel() == SO3X_MLP_PARAMS, "so3x: the fused training step is built for the ", SO3X_MLP_PARAMS, "-parameter skew-vector network");
  TORCH_CHECK(n > 0 && dout.numel() == n * 3 && t_used.numel() == n && loss.numel() >= 1 && (size_t)zstash.numel() >= so3x_mlp_stash_bytes(n) &&
              (!out.has_value() || out->numel() == n * 3), "so3x: train_net buffer sizes differ");
  dev(dout, "dout"); dev(loss, "loss"); dev(zstash, "zstash", at::kByte); dev(workspace, "workspace", at::kByte);
  ok(so3x_train_net(strm(x_t), F(dev(params, "params")), (int)T, F(dev(x_t, "x_t")), I64(dev(t_used, "t_used", at::kLong)), n, Fm(dout),
                    zstash.mutable_data_ptr(), Fm(loss), out.has_value() ? dev(*out, "out").mutable_data_ptr<float>() : nullptr,
                    rng_counter.has_value() ? dev(*rng_counter, "rng_counter", at::kLong).mutable_data_ptr<int64_t>() : nullptr,
                    workspace.mutable_data_ptr(), workspace.numel()),
     "train_net");
}
void train_bwd_partial(const Tensor& x_t, const Tensor& t, const Tensor& dout, const Tensor& zstash, int64_t T, Tensor& workspace) {
  GUARD(x_t);
  dev(workspace, "workspace", at::kByte);
  ok(so3x_train_bwd_partial(strm(x_t), F(dev(x_t, "x_t")), I64(dev(t, "t", at::kLong)), F(dev(dout, "dout")),
                            dev(zstash, "zstash", at::kByte).const_data_ptr(), x_t.numel() / 9, (int)T, workspace.mutable_data_ptr(), workspace.numel()),
     "train_bwd_partial");
}
void train_bwd_reduce(int64_t n, int64_t T, const optional<Tensor>& gscale, Tensor& grad, const Tensor& workspace) {
  GUARD(grad);
  TORCH_CHECK(grad.numel() == SO3X_MLP_PARAMS, "so3x: train_bwd_reduce writes the ", SO3X_MLP_PARAMS, "-parameter flat gradient");
  ok(so3x_train_bwd_reduce(strm(grad), n, (int)T, Fo(gscale, "grad_output"), dev(grad, "grad").mutable_data_ptr<float>(),
                           dev(workspace, "workspace", at::kByte).const_data_ptr(), workspace.numel()),
     "train_bwd_reduce");
}
void train_bwd_reduce_adam(int64_t n, int64_t T, const optional<Tensor>& gscale, Tensor& grad, const Tensor& workspace, Tensor& params,
                           Tensor& exp_avg, Tensor& exp_avg_sq, Tensor& step, double lr, double beta1, double beta2, double eps,
                           double weight_decay, double grad_scale) {
  GUARD(grad);
  const int64_t np = params.numel();
  TORCH_CHECK(np == SO3X_MLP_PARAMS && grad.numel() == np && exp_avg.numel() == np && exp_avg_sq.numel() == np && step.numel() >= 2,
              "so3x: train_bwd_reduce_adam works on the ", SO3X_MLP_PARAMS, "-parameter flat buffers");
  dev(params, "params"); dev(exp_avg, "exp_avg"); dev(exp_avg_sq, "exp_avg_sq"); dev(step, "step");
  ok(so3x_train_bwd_reduce_adam(strm(grad), n, (int)T, Fo(gscale, "grad_output"), dev(grad, "grad").mutable_data_ptr<float>(),
                                dev(workspace, "workspace", at::kByte).const_data_ptr(), workspace.numel(), Fm(params), Fm(exp_avg),
                                Fm(exp_avg_sq), Fm(step), (float)lr, (float)beta1, (float)beta2, (float)eps, (float)weight_decay,
                                (float)grad_scale),
     "train_bwd_reduce_adam");
}
void adam_step(Tensor& params, const Tensor& grad, Tensor& exp_avg, Tensor& exp_avg_sq, Tensor& step, double lr, double beta1, double beta2,
               double eps, double weight_decay, double grad_scale) {
  GUARD(params);
  const int64_t n = params.numel();
  TORCH_CHECK(grad.numel() == n && exp_avg.numel() == n && exp_avg_sq.numel() == n && step.numel() >= 2, "so3x: adam_step buffer sizes differ");
  dev(params, "params"); dev(exp_avg, "exp_avg"); dev(exp_avg_sq, "exp_avg_sq"); dev(step, "step");
  ok(so3x_adam_step(strm(params), Fm(params), F(dev(grad, "grad")), Fm(exp_avg), Fm(exp_avg_sq), Fm(step), n, (float)lr, (float)beta1,
                    (float)beta2, (float)eps, (float)weight_decay, (float)grad_scale),
     "adam_step");
}

// ------------------------------------------------------------------------------ widened rows
const void* Bo(const optional<Tensor>& x, const char* name) { return x.has_value() ? dev(*x, name, at::kByte).const_data_ptr() : nullptr; }

Tensor rotate_cloud(const Tensor& rot, const Tensor& cloud, int64_t cloud_stride, int64_t P) {
  GUARD(rot);
  Tensor out = f32_like(rot, with_tail(rot, 2, {P, 3}));
  ok(so3x_rotate_cloud(strm(rot), F(dev(rot, "x")), F(dev(cloud, "data")), cloud_stride, Fm(out), rot.numel() / 9, P), "rotate_cloud");
  return out;
}
// PlaneNet (models.py:185-210): forward (+ the stash the backward reads when want_stash) and backward
Tensor planenet_prepare(const Tensor& params, int64_t dim, int64_t heads, int64_t layers, int64_t ffn, int64_t precision) {
  GUARD(params);
  Tensor w = bytes(params, so3x_planenet_weights_bytes((int)dim, (int)heads, (int)layers, (int)ffn, (int)precision));
  ok(so3x_planenet_prepare(strm(params), F(dev(params, "params")), (int)dim, (int)heads, (int)layers, (int)ffn, (int)precision,
                           w.numel() ? w.mutable_data_ptr() : nullptr, w.numel()),
     "planenet_prepare");
  return w;
}
std::tuple<Tensor, Tensor, Tensor> planenet_fwd(const Tensor& params, const Tensor& x, const Tensor& t, int64_t dim, int64_t heads, int64_t layers,
                                                int64_t ffn, int64_t precision, bool want_stash, bool want_encoding, const optional<Tensor>& prepared,
                                                double dropout_p, int64_t seed, int64_t rng_offset) {
  GUARD(x);
  TORCH_CHECK(x.dim() == 3 && x.size(2) == 3, "so3x: x must be [B, P, 3]");
  const int64_t B = x.size(0), P = x.size(1);
  TORCH_CHECK(t.numel() == B, "so3x: t must hold one timestep per cloud");
  TORCH_CHECK(params.numel() == so3x_planenet_param_count((int)dim, (int)heads, (int)layers, (int)ffn), "so3x: params must hold ",
              so3x_planenet_param_count((int)dim, (int)heads, (int)layers, (int)ffn), " values");
  Tensor out = f32_like(x, {B, 3});
  Tensor enc = f32_like(x, {want_encoding ? B : 0, P, dim});
  const size_t wsb = so3x_planenet_workspace_bytes(B, P, (int)dim, (int)heads, (int)layers, (int)ffn, (int)precision);
  const size_t stb = want_stash ? so3x_planenet_stash_bytes(B, P, (int)dim, (int)heads, (int)layers, (int)ffn, (int)precision) : 0;
  TORCH_CHECK(B == 0 || wsb > 0, "so3x: planenet_fwd: this (dim, heads, layers, ffn, points, precision) is not supported");
  Tensor ws = bytes(x, wsb), stash = bytes(x, stb);
  ok(so3x_planenet_fwd(strm(x), F(dev(params, "params")), F(dev(x, "x")), I64(dev(t, "t", at::kLong)), Fm(out), want_encoding && B ? Fm(enc) : nullptr,
                       B, P, (int)dim, (int)heads, (int)layers, (int)ffn, (int)precision, want_stash && B ? stash.mutable_data_ptr() : nullptr,
                       ws.mutable_data_ptr(), ws.numel(), (prepared.has_value() && prepared->numel()) ? dev(*prepared, "prepared", at::kByte).const_data_ptr() : nullptr,
                       (float)dropout_p, (uint64_t)seed, (uint64_t)rng_offset),
     "planenet_fwd");
  return {out, stash, enc};
}
Tensor planenet_bwd(const Tensor& params, const Tensor& x, const Tensor& t, const Tensor& dout, const Tensor& stash, int64_t dim, int64_t heads,
                    int64_t layers, int64_t ffn, int64_t precision, double dropout_p, int64_t seed, int64_t rng_offset) {
  GUARD(x);
  const int64_t B = x.size(0), P = x.size(1);
  Tensor dparams = f32_like(x, {params.numel()});
  Tensor ws = bytes(x, so3x_planenet_workspace_bytes(B, P, (int)dim, (int)heads, (int)layers, (int)ffn, (int)precision));
  ok(so3x_planenet_bwd(strm(x), F(dev(params, "params")), F(dev(x, "x")), I64(dev(t, "t", at::kLong)), F(dev(dout, "dout")), Fm(dparams), B, P,
                       (int)dim, (int)heads, (int)layers, (int)ffn, (int)precision, dev(stash, "stash", at::kByte).const_data_ptr(),
                       ws.mutable_data_ptr(), ws.numel(), (float)dropout_p, (uint64_t)seed, (uint64_t)rng_offset),
     "planenet_bwd");
  return dparams;
}
// ProtNet (models.py:212-319): ragged complexes as concatenated ProtData fields + int64 [B + 1] offsets
std::tuple<Tensor, Tensor, Tensor, Tensor> protnet_fwd(const Tensor& params, const Tensor& rec_res, const Tensor& rec_pos, const Tensor& rec_ang,
                                                       const Tensor& rec_off, const Tensor& lig_res, const Tensor& lig_pos, const Tensor& lig_ang,
                                                       const Tensor& lig_off, const Tensor& t, int64_t max_len, int64_t dim, int64_t heads,
                                                       int64_t t_depth, int64_t c_depth, int64_t precision, bool want_stash, bool want_pool,
                                                       bool want_encoding, double dropout_p, int64_t seed, int64_t rng_offset) {
  GUARD(rec_pos);
  const int64_t B = t.numel(), nr = rec_pos.numel() / 3, nl = lig_pos.numel() / 3;
  TORCH_CHECK(rec_off.numel() == B + 1 && lig_off.numel() == B + 1, "so3x: protnet_fwd: offsets must hold B + 1 entries");
  TORCH_CHECK(rec_res.numel() == nr * 21 && rec_ang.numel() == nr * 9 && lig_res.numel() == nl * 21 && lig_ang.numel() == nl * 9,
              "so3x: protnet_fwd: residues [n, 21], positions [n, 3], angles [n, 3, 3] per chain kind");
  const int64_t np = so3x_protnet_param_count((int)dim, (int)heads, (int)t_depth, (int)c_depth);
  TORCH_CHECK(np > 0, "so3x: protnet_fwd: no ProtNet with this (dim, heads, t_depth, c_depth)");
  TORCH_CHECK(params.numel() == np, "so3x: params must hold ", np, " values");
  Tensor out = f32_like(rec_pos, {B, 6});
  Tensor pool = f32_like(rec_pos, {want_pool ? B : 0, 3 * dim + 6});
  Tensor enc = f32_like(rec_pos, {want_encoding ? 2 * B : 0, max_len, dim});
  const size_t wsb = so3x_protnet_workspace_bytes(B, max_len, nr, nl, (int)dim, (int)heads, (int)t_depth, (int)c_depth, (int)precision);
  const size_t stb = want_stash ? so3x_protnet_stash_bytes(B, max_len, (int)dim, (int)heads, (int)t_depth, (int)c_depth, (int)precision) : 0;
  TORCH_CHECK(B == 0 || wsb > 0, "so3x: protnet_fwd: this (dim, heads, t_depth, c_depth, max_len, precision) is not supported");
  TORCH_CHECK(!want_stash || B == 0 || stb > 0, "so3x: protnet_fwd: no training stash for this precision");
  Tensor ws = bytes(rec_pos, wsb), stash = bytes(rec_pos, stb);
  ok(so3x_protnet_fwd(strm(rec_pos), F(dev(params, "params")), F(dev(rec_res, "rec_res")), F(dev(rec_pos, "rec_pos")), F(dev(rec_ang, "rec_ang")),
                      I64(dev(rec_off, "rec_off", at::kLong)), nr, F(dev(lig_res, "lig_res")), F(dev(lig_pos, "lig_pos")), F(dev(lig_ang, "lig_ang")),
                      I64(dev(lig_off, "lig_off", at::kLong)), nl, I64(dev(t, "t", at::kLong)), Fm(out), want_pool && B ? Fm(pool) : nullptr,
                      want_encoding && B ? Fm(enc) : nullptr, B, max_len, (int)dim, (int)heads, (int)t_depth, (int)c_depth, (int)precision,
                      want_stash && B ? stash.mutable_data_ptr() : nullptr, ws.mutable_data_ptr(), ws.numel(), (float)dropout_p, (uint64_t)seed,
                      (uint64_t)rng_offset),
     "protnet_fwd");
  return {out, stash, pool, enc};
}
Tensor protnet_bwd(const Tensor& params, const Tensor& dout, const Tensor& stash, int64_t max_len, int64_t dim, int64_t heads, int64_t t_depth,
                   int64_t c_depth, int64_t precision, double dropout_p, int64_t seed, int64_t rng_offset) {
  GUARD(dout);
  const int64_t B = dout.numel() / 6;
  Tensor dparams = f32_like(dout, {params.numel()});
  Tensor ws = bytes(dout, so3x_protnet_workspace_bytes(B, max_len, 0, 0, (int)dim, (int)heads, (int)t_depth, (int)c_depth, (int)precision));
  ok(so3x_protnet_bwd(strm(dout), F(dev(params, "params")), F(dev(dout, "dout")), Fm(dparams), B, max_len, (int)dim, (int)heads, (int)t_depth,
                      (int)c_depth, (int)precision, dev(stash, "stash", at::kByte).const_data_ptr(), ws.mutable_data_ptr(), ws.numel(), (float)dropout_p,
                      (uint64_t)seed, (uint64_t)rng_offset),
     "protnet_bwd");
  return dparams;
}
// wide residual score network (so3_lock_train.py:11-59)
Tensor resnet_fwd(const Tensor& params, const Tensor& x, const Tensor& t, int64_t t_stride, int64_t n_out, int64_t precision, int64_t t_table) {
  GUARD(x);
  Tensor out = f32_like(x, with_tail(x, 2, {n_out}));
  Tensor ws = bytes(x, so3x_resnet_workspace_bytes((int)precision, (int)t_table));
  ok(so3x_resnet_fwd(strm(x), F(dev(params, "params")), F(dev(x, "x")), I64(dev(t, "t", at::kLong)), t_stride, Fm(out), x.numel() / 9, (int)n_out,
                     (int)precision, (int)t_table, ws.mutable_data_ptr(), ws.numel()),
     "resnet_fwd");
  return out;
}
std::tuple<Tensor, Tensor> resnet_fwd_stash(const Tensor& params, const Tensor& x, const Tensor& t, int64_t t_stride, int64_t n_out,
                                            int64_t precision, int64_t t_table) {
  GUARD(x);
  const int64_t n = x.numel() / 9;
  Tensor out = f32_like(x, with_tail(x, 2, {n_out}));
  Tensor stash = bytes(x, so3x_resnet_stash_bytes(n, (int)precision)), ws = bytes(x, so3x_resnet_workspace_bytes((int)precision, (int)t_table));
  ok(so3x_resnet_fwd_stash(strm(x), F(dev(params, "params")), F(dev(x, "x")), I64(dev(t, "t", at::kLong)), t_stride, Fm(out),
                           stash.mutable_data_ptr(), n, (int)n_out, (int)precision, (int)t_table, ws.mutable_data_ptr(), ws.numel()),
     "resnet_fwd_stash");
  return {out, stash};
}
Tensor resnet_bwd(const Tensor& params, const Tensor& x, const Tensor& t, int64_t t_stride, const Tensor& dout, int64_t n_out, int64_t precision,
                  int64_t t_table, const optional<Tensor>& stash) {
  GUARD(x);
  const int64_t n = x.numel() / 9;
  Tensor dparams = f32_like(x, {params.numel()});
  Tensor ws = bytes(x, so3x_resnet_train_workspace_bytes(n, (int)precision, (int)t_table));
  ok(so3x_resnet_bwd(strm(x), F(dev(params, "params")), F(dev(x, "x")), I64(dev(t, "t", at::kLong)), t_stride, F(dev(dout, "dout")), Fm(dparams), n,
                     (int)n_out, (int)precision, (int)t_table, Bo(stash, "stash"), ws.mutable_data_ptr(), ws.numel()),
     "resnet_bwd");
  return dparams;
}
void resnet_chain_into(const Tensor& params, const Tensor& sched, const Tensor& trap_p, const optional<Tensor>& guide_p, const Tensor& x,
                       const Tensor& out, int64_t t_start, int64_t n_steps, const optional<Tensor>& axes, const optional<Tensor>& unif,
                       int64_t seed, int64_t rng_offset, int64_t index_base, int64_t precision) {
  const int T = (int)dev(sched, "sched").size(1);
  TORCH_CHECK(out.numel() == x.numel() && out.device() == x.device(), "so3x: out must match x");
  Tensor ws = bytes(x, so3x_resnet_workspace_bytes((int)precision, T));
  ok(so3x_resnet_p_sample_chain(strm(x), F(dev(params, "params")), F(sched), T, F(dev(trap_p, "trap_p")), Guide(guide_p), F(dev(x, "x")),
                                Fm(const_cast<Tensor&>(dev(out, "out"))), (int)t_start, (int)n_steps, Fo(axes, "axes"), Fo(unif, "unif"),
                                (uint64_t)seed, (uint64_t)rng_offset, index_base, x.numel() / 9, (int)precision, ws.mutable_data_ptr(), ws.numel()),
     "resnet_p_sample_chain");
}
Tensor resnet_p_sample_chain(const Tensor& params, const Tensor& sched, const Tensor& trap_p, const optional<Tensor>& guide_p, const Tensor& x,
                             int64_t t_start, int64_t n_steps, const optional<Tensor>& axes, const optional<Tensor>& unif, int64_t seed,
                             int64_t rng_offset, int64_t index_base, int64_t precision) {
  GUARD(x);
  Tensor out = at::empty_like(x);
  resnet_chain_into(params, sched, trap_p, guide_p, x, out, t_start, n_steps, axes, unif, seed, rng_offset, index_base, precision);
  return out;
}
void resnet_p_sample_chain_out(const Tensor& params, const Tensor& sched, const Tensor& trap_p, const optional<Tensor>& guide_p, const Tensor& x,
                               int64_t t_start, int64_t n_steps, const optional<Tensor>& axes, const optional<Tensor>& unif, int64_t seed,
                               int64_t rng_offset, int64_t index_base, int64_t precision, Tensor& out) {
  GUARD(x);
  resnet_chain_into(params, sched, trap_p, guide_p, x, out, t_start, n_steps, axes, unif, seed, rng_offset, index_base, precision);
}
// SE(3) layer
std::tuple<Tensor, Tensor, Tensor, Tensor> se3_q_sample_target(const Tensor& sched, const Tensor& trap_q, const optional<Tensor>& guide_q,
                                                               double shift_scale, const Tensor& x0_rot, const Tensor& x0_shift, const Tensor& t,
                                                               bool quirk_col0, const optional<Tensor>& axes, const optional<Tensor>& unif,
                                                               const optional<Tensor>& znorm, int64_t seed, int64_t rng_offset, int64_t index_base,
                                                               bool want_targets) {
  GUARD(x0_rot);
  const int64_t n = x0_rot.numel() / 9;
  const int T = (int)dev(sched, "sched").size(1);
  Tensor xt_rot = at::empty_like(dev(x0_rot, "x_start.rot")), xt_shift = f32_like(x0_rot, {n, 3});
  Tensor tg_rot = f32_like(x0_rot, {want_targets ? n : 0, 3}), tg_shift = f32_like(x0_rot, {want_targets ? n : 0, 3});
  ok(so3x_se3_q_sample_target(strm(x0_rot), F(sched), T, F(dev(trap_q, "trap_q")), Guide(guide_q), (float)shift_scale, F(x0_rot),
                              F(dev(x0_shift, "x_start.shift")), I64(dev(t, "t", at::kLong)), quirk_col0 ? 1 : 0, Fo(axes, "axes"), Fo(unif, "unif"),
                              Fo(znorm, "znorm"), (uint64_t)seed, (uint64_t)rng_offset, index_base, Fm(xt_rot), Fm(xt_shift),
                              want_targets ? Fm(tg_rot) : nullptr, want_targets ? Fm(tg_shift) : nullptr, n),
     "se3_q_sample_target");
  return {xt_rot, xt_shift, tg_rot, tg_shift};
}
std::tuple<Tensor, Tensor> se3_p_mean(const Tensor& sched, const Tensor& x_rot, const Tensor& x_shift, const Tensor& v_rot, const Tensor& v_shift,
                                      int64_t t) {
  GUARD(x_rot);
  const int T = (int)dev(sched, "sched").size(1);
  Tensor mean_rot = at::empty_like(dev(x_rot, "x.rot")), mean_shift = at::empty_like(dev(x_shift, "x.shift"));
  ok(so3x_se3_p_mean(strm(x_rot), F(sched), T, F(x_rot), F(x_shift), F(dev(v_rot, "noise.rot_g")), F(dev(v_shift, "noise.shift_g")), (int)t,
                     Fm(mean_rot), Fm(mean_shift), x_rot.numel() / 9),
     "se3_p_mean");
  return {mean_rot, mean_shift};
}
std::tuple<Tensor, Tensor> se3_p_noise(const Tensor& trap_row, double sigma, double shift_scale, const Tensor& mean_rot, const Tensor& mean_shift,
                                       const optional<Tensor>& axes, const optional<Tensor>& unif, const optional<Tensor>& znorm, int64_t seed,
                                       int64_t rng_offset, int64_t index_base, bool shared_rot) {
  GUARD(mean_rot);
  Tensor out_rot = at::empty_like(dev(mean_rot, "mean.rot")), out_shift = at::empty_like(dev(mean_shift, "mean.shift"));
  ok(so3x_se3_p_noise(strm(mean_rot), F(dev(trap_row, "trap_row")), (float)sigma, (float)shift_scale, F(mean_rot), F(mean_shift), Fo(axes, "axes"),
                      Fo(unif, "unif"), Fo(znorm, "znorm"), (uint64_t)seed, (uint64_t)rng_offset, index_base, shared_rot ? 1 : 0, Fm(out_rot),
                      Fm(out_shift), mean_rot.numel() / 9),
     "se3_p_noise");
  return {out_rot, out_shift};
}
std::tuple<Tensor, Tensor> rigid_move(const Tensor& rot, const Tensor& shift, const Tensor& pos, const optional<Tensor>& frames) {
  GUARD(rot);
  const int64_t S = rot.numel() / 9, L = S ? pos.numel() / (3 * S) : 0;
  Tensor out_pos = at::empty_like(dev(pos, "positions"));
  Tensor out_fr = frames.has_value() ? at::empty_like(dev(*frames, "angles")) : f32_like(rot, {0, 3, 3});
  ok(so3x_rigid_move(strm(rot), F(dev(rot, "transf.rot")), F(dev(shift, "transf.shift")), F(pos), Fo(frames, "angles"), Fm(out_pos),
                     frames.has_value() ? Fm(out_fr) : nullptr, S, L),
     "rigid_move");
  return {out_pos, out_fr};
}
std::tuple<Tensor, Tensor> rigid_move_ragged(const Tensor& rot, const Tensor& shift, const Tensor& pos, const optional<Tensor>& frames, const Tensor& off) {
  GUARD(rot);
  const int64_t S = rot.numel() / 9;
  TORCH_CHECK(off.numel() == S + 1, "so3x: rigid_move_ragged: offsets must hold one entry per structure + 1");
  Tensor out_pos = at::empty_like(dev(pos, "positions"));
  Tensor out_fr = frames.has_value() ? at::empty_like(dev(*frames, "angles")) : f32_like(rot, {0, 3, 3});
  ok(so3x_rigid_move_ragged(strm(rot), F(dev(rot, "transf.rot")), F(dev(shift, "transf.shift")), F(pos), Fo(frames, "angles"),
                            I64(dev(off, "offsets", at::kLong)), Fm(out_pos), frames.has_value() ? Fm(out_fr) : nullptr, S),
     "rigid_move_ragged");
  return {out_pos, out_fr};
}
// statistics
Tensor kernel_sum(const Tensor& X, const Tensor& Y, int64_t kind, double scale) {
  GUARD(X);
  const int64_t nx = X.numel() / 9, ny = Y.numel() / 9;
  Tensor out = f32_like(X, {1}), ws = bytes(X, so3x_kernel_sum_workspace_bytes(nx, ny));
  ok(so3x_kernel_sum(strm(X), F(dev(X, "X")), nx, F(dev(Y, "Y")), ny, (int)kind, (float)scale, Fm(out), ws.mutable_data_ptr(), ws.numel()), "kernel_sum");
  return out;
}
// rotation-matrix head and the prevstep objective
Tensor six2rmat(const Tensor& x6) {
  GUARD(x6);
  Tensor out = f32_like(x6, with_tail(x6, 1, {3, 3}));
  ok(so3x_six2rmat(strm(x6), F(dev(x6, "x")), Fm(out), x6.numel() / 6), "six2rmat");
  return out;
}
Tensor six2rmat_bwd(const Tensor& x6, const Tensor& dR) {
  GUARD(x6);
  Tensor dx = at::empty_like(dev(x6, "x"));
  ok(so3x_six2rmat_bwd(strm(x6), F(x6), F(dev(dR, "grad")), Fm(dx), x6.numel() / 6), "six2rmat_bwd");
  return dx;
}
Tensor log_rmat_bwd(const Tensor& R, const Tensor& dlog) {
  GUARD(R);
  Tensor dR = at::empty_like(dev(R, "r_mat"));
  ok(so3x_log_rmat_bwd(strm(R), F(R), F(dev(dlog, "grad")), Fm(dR), R.numel() / 9), "log_rmat_bwd");
  return dR;
}
std::tuple<Tensor, Tensor> rmat_dist_bwd(const Tensor& a, const Tensor& b, const Tensor& ddist) {
  GUARD(a);
  Tensor da = at::empty_like(dev(a, "input")), db = at::empty_like(dev(b, "target"));
  ok(so3x_rmat_dist_bwd(strm(a), F(a), F(b), F(dev(ddist, "grad")), Fm(da), Fm(db), a.numel() / 9), "rmat_dist_bwd");
  return {da, db};
}
// -> (loss[1], d loss / d x_recon, step): the last two on request (else empty)
std::tuple<Tensor, Tensor, Tensor> prevstep_loss(const Tensor& sched, const Tensor& x_recon, const Tensor& x_start, const Tensor& x_noisy,
                                                 const Tensor& t, int64_t t_stride, bool want_dx, bool want_step) {
  GUARD(x_recon);
  const int64_t n = x_recon.numel() / 9;
  const int T = (int)dev(sched, "sched").size(1);
  Tensor loss = f32_like(x_recon, {1}), ws = bytes(x_recon, so3x_prevstep_workspace_bytes(n));
  Tensor dx = want_dx ? at::empty_like(x_recon) : f32_like(x_recon, {0, 3, 3}), step = want_step ? at::empty_like(x_recon) : f32_like(x_recon, {0, 3, 3});
  ok(so3x_prevstep_loss(strm(x_recon), F(sched), T, F(dev(x_recon, "x_recon")), F(dev(x_start, "x_start")), F(dev(x_noisy, "x_noisy")),
                        I64(dev(t, "t", at::kLong)), t_stride, n, Fm(loss), want_dx ? Fm(dx) : nullptr, want_step ? Fm(step) : nullptr,
                        ws.mutable_data_ptr(), ws.numel()),
     "prevstep_loss");
  return {loss, dx, step};
}
std::tuple<Tensor, Tensor> prevstep_loss6(const Tensor& sched, const Tensor& out6, const Tensor& x_start, const Tensor& x_noisy, const Tensor& t,
                                          int64_t t_stride) {
  GUARD(out6);
  const int64_t n = out6.numel() / 6;
  const int T = (int)dev(sched, "sched").size(1);
  Tensor loss = f32_like(out6, {1}), d6 = at::empty_like(dev(out6, "out6")), ws = bytes(out6, so3x_prevstep_workspace_bytes(n));
  ok(so3x_prevstep_loss6(strm(out6), F(sched), T, F(out6), F(dev(x_start, "x_start")), F(dev(x_noisy, "x_noisy")), I64(dev(t, "t", at::kLong)), t_stride,
                         n, Fm(loss), Fm(d6), ws.mutable_data_ptr(), ws.numel()),
     "prevstep_loss6");
  return {loss, d6};
}
Tensor mse_loss(const Tensor& a, const Tensor& b) {
  GUARD(a);
  TORCH_CHECK(a.numel() == b.numel(), "so3x: mse_loss needs equal shapes");
  Tensor loss = f32_like(a, {1}), ws = bytes(a, so3x_mse_workspace_bytes(a.numel()));
  ok(so3x_mse_loss(strm(a), F(dev(a, "input")), F(dev(b, "target")), a.numel(), Fm(loss), ws.mutable_data_ptr(), ws.numel()), "mse_loss");
  return loss;
}
Tensor mse_grad(const Tensor& a, const Tensor& b, const Tensor& gscale) {
  GUARD(a);
  Tensor ga = at::empty_like(dev(a, "input"));
  ok(so3x_mse_grad(strm(a), F(a), F(dev(b, "target")), a.numel(), F(dev(gscale, "grad")), Fm(ga)), "mse_grad");
  return ga;
}

}  // namespace

TORCH_LIBRARY(so3x, m) {
  m.def("quat_to_rmat(Tensor q) -> Tensor");
  m.def("log_rmat(Tensor R) -> Tensor");
  m.def("log_rmat_vec(Tensor R) -> Tensor");
  m.def("exp_skewvec(Tensor v) -> Tensor");
  m.def("orthogonalise(Tensor M) -> Tensor");
  m.def("so3_scale(Tensor R, Tensor k, int k_stride) -> Tensor");
  m.def("aa_to_rmat(Tensor axis, Tensor ang) -> Tensor");
  m.def("rmat_to_aa(Tensor R) -> (Tensor, Tensor)");
  m.def("so3_lerp(Tensor a, int a_stride, Tensor b, Tensor w, int w_stride) -> Tensor");
  m.def("rmat_dist(Tensor a, Tensor b) -> Tensor");
  m.def("rmul(Tensor a, int a_stride, Tensor b, int b_stride, bool transpose_b) -> Tensor");
  m.def("igso3_eps_ft(Tensor omega, Tensor eps, int eps_stride) -> Tensor");
  m.def("igso3_build_tables(Tensor eps) -> Tensor");
  m.def("igso3_build_guide(Tensor trap) -> Tensor");
  m.def("igso3_sample(Tensor trap, Tensor? guide, Tensor? row_idx, int row_const, bool quirk_col0, Tensor? axes, Tensor? unif, int seed, "
        "int rng_offset, int index_base, Tensor? mean, int n, bool want_angle, bool want_axis) -> (Tensor, Tensor, Tensor)");
  m.def("igso3_logprob_score(Tensor R, Tensor eps, int eps_stride, bool want_score, bool want_grad) -> (Tensor, Tensor, Tensor)");
  m.def("mlp_fwd(Tensor params, Tensor x, Tensor t, int t_stride, int precision, int t_table) -> Tensor");
  m.def("mlp_fwd_stash(Tensor params, Tensor x, Tensor t, int t_stride, int t_table) -> (Tensor, Tensor)");
  m.def("mlp_bwd(Tensor params, Tensor x, Tensor t, int t_stride, Tensor dout, int precision, int t_table, Tensor? zstash) -> Tensor");
  m.def("q_sample_target(Tensor sched, Tensor? trap_q, Tensor? guide_q, Tensor x0, Tensor t, bool quirk_col0, Tensor? noise, Tensor? axes, "
        "Tensor? unif, int seed, int rng_offset, Tensor? rng_offset_dev, int index_base, bool want_x_t, bool want_target, bool want_noise) "
        "-> (Tensor, Tensor, Tensor)");
  m.def("p_mean(Tensor sched, Tensor x, Tensor v, Tensor? t, int t_stride, int t_const, bool want_x0hat) -> (Tensor, Tensor)");
  m.def("p_sample_prepare(Tensor params, Tensor sched, Tensor trap_p, Tensor? guide_p, int precision) -> Tensor");
  m.def("resnet_p_sample_prepare(Tensor params, int T, int precision) -> Tensor");
  m.def("resnet_p_sample_prepared(Tensor workspace, Tensor sched, Tensor trap_p, Tensor? guide_p, Tensor x, int t_start, Tensor? t_dev, int n_steps, Tensor? axes, Tensor? unif, int seed, int rng_offset, int index_base, int precision) -> Tensor");
  m.def("resnet_p_sample_prepared_out(Tensor workspace, Tensor sched, Tensor trap_p, Tensor? guide_p, Tensor x, int t_start, Tensor? t_dev, int n_steps, Tensor? axes, Tensor? unif, int seed, int rng_offset, int index_base, int precision, Tensor(b!) out) -> ()");
  m.def("p_sample_prepared(Tensor(a!) workspace, Tensor sched, Tensor trap_p, Tensor? guide_p, Tensor x, int t_start, Tensor? t_dev, int n_steps, Tensor? axes, Tensor? unif, int seed, int rng_offset, int index_base, int precision) -> Tensor");
  m.def("p_sample_prepared_out(Tensor(a!) workspace, Tensor sched, Tensor trap_p, Tensor? guide_p, Tensor x, int t_start, Tensor? t_dev, int n_steps, Tensor? axes, Tensor? unif, int seed, int rng_offset, int index_base, int precision, Tensor(b!) out) -> ()");
  m.def("p_sample_chain(Tensor params, Tensor sched, Tensor trap_p, Tensor? guide_p, Tensor x, int t_start, int n_steps, Tensor? axes, "
        "Tensor? unif, int seed, int rng_offset, int index_base, int precision) -> Tensor");
  m.def("p_sample_chain_out(Tensor params, Tensor sched, Tensor trap_p, Tensor? guide_p, Tensor x, int t_start, int n_steps, Tensor? axes, "
        "Tensor? unif, int seed, int rng_offset, int index_base, int precision, Tensor(a!) out) -> ()");
  m.def("train_fwd(Tensor params, Tensor sched, Tensor trap_q, Tensor? guide_q, Tensor x0, Tensor? t, bool quirk_col0, Tensor? axes, "
        "Tensor? unif, int seed, int rng_offset, Tensor(a!)? rng_counter, int index_base, bool want_out) "
        "-> (Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor)");
  m.def("train_bwd(Tensor x_t, Tensor t, Tensor dout, Tensor zstash, Tensor(a!) workspace, int T, Tensor? gscale, int n_params) -> Tensor");
  m.def("train_fused(Tensor params, Tensor sched, Tensor trap_q, Tensor? guide_q, Tensor x0, Tensor? t, bool quirk_col0, Tensor? axes, "
        "Tensor? unif, int seed, int rng_offset, Tensor(a!)? rng_counter, int index_base, Tensor(b!) loss, Tensor(c!)? t_used, "
        "Tensor(d!)? x_t, Tensor(e!)? out, Tensor(f!) workspace) -> ()");
  m.def("train_noise(Tensor sched, Tensor trap_q, Tensor? guide_q, Tensor x0, Tensor? t, bool quirk_col0, Tensor? axes, Tensor? unif, int seed, "
        "int rng_offset, Tensor? rng_counter, int index_base, Tensor(a!) x_t, Tensor(b!) t_used, Tensor(c!) workspace) -> ()");
  m.def("train_net(Tensor params, int T, Tensor x_t, Tensor t_used, Tensor(a!) dout, Tensor(b!) zstash, Tensor(c!) loss, Tensor(d!)? out, "
        "Tensor(e!)? rng_counter, Tensor(f!) workspace) -> ()");
  m.def("train_bwd_partial(Tensor x_t, Tensor t, Tensor dout, Tensor zstash, int T, Tensor(a!) workspace) -> ()");
  m.def("train_bwd_reduce(int n, int T, Tensor? gscale, Tensor(a!) grad, Tensor workspace) -> ()");
  m.def("train_bwd_reduce_adam(int n, int T, Tensor? gscale, Tensor(a!) grad, Tensor workspace, Tensor(b!) params, Tensor(c!) exp_avg, "
        "Tensor(d!) exp_avg_sq, Tensor(e!) step, float lr, float beta1, float beta2, float eps, float weight_decay, float grad_scale) -> ()");
  m.def("adam_step(Tensor(a!) params, Tensor grad, Tensor(b!) exp_avg, Tensor(c!) exp_avg_sq, Tensor(d!) step, float lr, float beta1, "
        "float beta2, float eps, float weight_decay, float grad_scale) -> ()");
  m.def("rotate_cloud(Tensor rot, Tensor cloud, int cloud_stride, int P) -> Tensor");
  m.def("resnet_fwd(Tensor params, Tensor x, Tensor t, int t_stride, int n_out, int precision, int t_table) -> Tensor");
  m.def("planenet_prepare(Tensor params, int dim, int heads, int layers, int ffn, int precision) -> Tensor");
  m.def("planenet_fwd(Tensor params, Tensor x, Tensor t, int dim, int heads, int layers, int ffn, int precision, bool want_stash, bool want_encoding, Tensor? prepared, float dropout_p, int seed, int rng_offset) -> (Tensor, Tensor, Tensor)");
  m.def("protnet_fwd(Tensor params, Tensor rec_res, Tensor rec_pos, Tensor rec_ang, Tensor rec_off, Tensor lig_res, Tensor lig_pos, Tensor lig_ang, Tensor lig_off, Tensor t, int max_len, int dim, int heads, int t_depth, int c_depth, int precision, bool want_stash, bool want_pool, bool want_encoding, float dropout_p, int seed, int rng_offset) -> (Tensor, Tensor, Tensor, Tensor)");
  m.def("protnet_bwd(Tensor params, Tensor dout, Tensor stash, int max_len, int dim, int heads, int t_depth, int c_depth, int precision, float dropout_p, int seed, int rng_offset) -> Tensor");
  m.def("planenet_bwd(Tensor params, Tensor x, Tensor t, Tensor dout, Tensor stash, int dim, int heads, int layers, int ffn, int precision, float dropout_p, int seed, int rng_offset) -> Tensor");
  m.def("resnet_fwd_stash(Tensor params, Tensor x, Tensor t, int t_stride, int n_out, int precision, int t_table) -> (Tensor, Tensor)");
  m.def("resnet_bwd(Tensor params, Tensor x, Tensor t, int t_stride, Tensor dout, int n_out, int precision, int t_table, Tensor? stash) -> Tensor");
  m.def("resnet_p_sample_chain(Tensor params, Tensor sched, Tensor trap_p, Tensor? guide_p, Tensor x, int t_start, int n_steps, Tensor? axes, "
        "Tensor? unif, int seed, int rng_offset, int index_base, int precision) -> Tensor");
  m.def("resnet_p_sample_chain_out(Tensor params, Tensor sched, Tensor trap_p, Tensor? guide_p, Tensor x, int t_start, int n_steps, Tensor? axes, "
        "Tensor? unif, int seed, int rng_offset, int index_base, int precision, Tensor(a!) out) -> ()");
  m.def("se3_q_sample_target(Tensor sched, Tensor trap_q, Tensor? guide_q, float shift_scale, Tensor x0_rot, Tensor x0_shift, Tensor t, "
        "bool quirk_col0, Tensor? axes, Tensor? unif, Tensor? znorm, int seed, int rng_offset, int index_base, bool want_targets) "
        "-> (Tensor, Tensor, Tensor, Tensor)");
  m.def("se3_p_mean(Tensor sched, Tensor x_rot, Tensor x_shift, Tensor v_rot, Tensor v_shift, int t) -> (Tensor, Tensor)");
  m.def("se3_p_noise(Tensor trap_row, float sigma, float shift_scale, Tensor mean_rot, Tensor mean_shift, Tensor? axes, Tensor? unif, "
        "Tensor? znorm, int seed, int rng_offset, int index_base, bool shared_rot) -> (Tensor, Tensor)");
  m.def("rigid_move(Tensor rot, Tensor shift, Tensor pos, Tensor? frames) -> (Tensor, Tensor)");
  m.def("rigid_move_ragged(Tensor rot, Tensor shift, Tensor pos, Tensor? frames, Tensor off) -> (Tensor, Tensor)");
  m.def("kernel_sum(Tensor X, Tensor Y, int kind, float scale) -> Tensor");
  m.def("six2rmat(Tensor x6) -> Tensor");
  m.def("six2rmat_bwd(Tensor x6, Tensor dR) -> Tensor");
  m.def("log_rmat_bwd(Tensor R, Tensor dlog) -> Tensor");
  m.def("rmat_dist_bwd(Tensor a, Tensor b, Tensor ddist) -> (Tensor, Tensor)");
  m.def("prevstep_loss(Tensor sched, Tensor x_recon, Tensor x_start, Tensor x_noisy, Tensor t, int t_stride, bool want_dx, bool want_step) "
        "-> (Tensor, Tensor, Tensor)");
  m.def("prevstep_loss6(Tensor sched, Tensor out6, Tensor x_start, Tensor x_noisy, Tensor t, int t_stride) -> (Tensor, Tensor)");
  m.def("mse_loss(Tensor a, Tensor b) -> Tensor");
  m.def("mse_grad(Tensor a, Tensor b, Tensor gscale) -> Tensor");
}

TORCH_LIBRARY_IMPL(so3x, CUDA, m) {
  m.impl("quat_to_rmat", quat_to_rmat);
  m.impl("log_rmat", log_rmat);
  m.impl("log_rmat_vec", log_rmat_vec);
  m.impl("exp_skewvec", exp_skewvec);
  m.impl("orthogonalise", orthogonalise);
  m.impl("so3_scale", so3_scale);
  m.impl("aa_to_rmat", aa_to_rmat);
  m.impl("rmat_to_aa", rmat_to_aa);
  m.impl("so3_lerp", so3_lerp);
  m.impl("rmat_dist", rmat_dist);
  m.impl("rmul", rmul);
  m.impl("igso3_eps_ft", igso3_eps_ft);
  m.impl("igso3_build_tables", igso3_build_tables);
  m.impl("igso3_build_guide", igso3_build_guide);
  m.impl("igso3_sample", igso3_sample);
  m.impl("igso3_logprob_score", igso3_logprob_score);
  m.impl("mlp_fwd", mlp_fwd);
  m.impl("mlp_fwd_stash", mlp_fwd_stash);
  m.impl("mlp_bwd", mlp_bwd);
  m.impl("q_sample_target", q_sample_target);
  m.impl("p_mean", p_mean);
  m.impl("p_sample_prepare", p_sample_prepare);
  m.impl("resnet_p_sample_prepare", resnet_p_sample_prepare);
  m.impl("resnet_p_sample_prepared", resnet_p_sample_prepared);
  m.impl("resnet_p_sample_prepared_out", resnet_p_sample_prepared_out);
  m.impl("p_sample_prepared", p_sample_prepared);
  m.impl("p_sample_prepared_out", p_sample_prepared_out);
  m.impl("p_sample_chain", p_sample_chain);
  m.impl("p_sample_chain_out", p_sample_chain_out);
  m.impl("train_fwd", train_fwd);
  m.impl("train_bwd", train_bwd);
  m.impl("train_fused", train_fused);
  m.impl("train_noise", train_noise);
  m.impl("train_net", train_net);
  m.impl("train_bwd_partial", train_bwd_partial);
  m.impl("train_bwd_reduce", train_bwd_reduce);
  m.impl("train_bwd_reduce_adam", train_bwd_reduce_adam);
  m.impl("adam_step", adam_step);
  m.impl("rotate_cloud", rotate_cloud);
  m.impl("resnet_fwd", resnet_fwd);
  m.impl("planenet_prepare", planenet_prepare);
  m.impl("planenet_fwd", planenet_fwd);
  m.impl("planenet_bwd", planenet_bwd);
  m.impl("protnet_fwd", protnet_fwd);
  m.impl("protnet_bwd", protnet_bwd);
  m.impl("resnet_fwd_stash", resnet_fwd_stash);
  m.impl("resnet_bwd", resnet_bwd);
  m.impl("resnet_p_sample_chain", resnet_p_sample_chain);
  m.impl("resnet_p_sample_chain_out", resnet_p_sample_chain_out);
  m.impl("se3_q_sample_target", se3_q_sample_target);
  m.impl("se3_p_mean", se3_p_mean);
  m.impl("se3_p_noise", se3_p_noise);
  m.impl("rigid_move", rigid_move);
  m.impl("rigid_move_ragged", rigid_move_ragged);
  m.impl("kernel_sum", kernel_sum);
  m.impl("six2rmat", six2rmat);
  m.impl("six2rmat_bwd", six2rmat_bwd);
  m.impl("log_rmat_bwd", log_rmat_bwd);
  m.impl("rmat_dist_bwd", rmat_dist_bwd);
  m.impl("prevstep_loss", prevstep_loss);
  m.impl("prevstep_loss6", prevstep_loss6);
  m.impl("mse_loss", mse_loss);
  m.impl("mse_grad", mse_grad);
}
