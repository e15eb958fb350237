/*
 * so3x.h -- C ABI of libso3x.so, the MI355X-native (gfx950) backend of the SO(3)
 * diffusion hot path of qazwsxal/diffusion-extensions.
 *
 * The reference is pure Python/PyTorch with no FFI layer of its own; each entry
 * point below names the reference function (file:line under the reference
 * checkout) whose work it replaces.  The Python host classes that keep the
 * reference's public names (SO3Diffusion, IsotropicGaussianSO3, RotPredict,
 * util.*) bind these symbols with ctypes -- INTEGRATION.md shows the stub.
 *
 * Conventions
 *  - extern "C", plain pointers and sizes, no torch / C++ types.
 *  - every function returns int: 0 = SO3X_OK, > 0 = a hipError_t from the
 *    launch, < 0 = an SO3X_ERR_* argument error.  Nothing throws.
 *  - device pointers unless a parameter is marked [host]; fp32, contiguous,
 *    rotations row-major [n][3][3] (36 B per sample, AoS as the reference's
 *    tensors); int64 timesteps.
 *  - no hidden allocation and no host synchronisation: the caller owns all
 *    buffers, passes the workspace, and names the HIP stream (hipStream_t as
 *    void*).  All launches are graph-capturable.  Reentrant, no mutable globals.
 *  - "quirk" flags reproduce documented reference behaviour bit-faithfully and
 *    can be switched off.
 */
#ifndef SO3X_H
#define SO3X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SO3X_ABI_VERSION 8

#define SO3X_OK 0
#define SO3X_ERR_INVALID_ARG (-1)
#define SO3X_ERR_WORKSPACE (-2)
#define SO3X_ERR_UNSUPPORTED (-3)

/* MLP operand precision: rotation state and rotation math are always fp32. */
#define SO3X_PREC_F32 0  /* exact-fp32 MFMA (v_mfma_f32_32x32x2_f32)                 */
#define SO3X_PREC_BF16 1 /* bf16 operands, fp32 accumulate (v_mfma_f32_32x32x16_bf16) */
#define SO3X_PREC_F16 2   /* so3x_p_sample_chain ONLY (round 4): the bf16 path with IEEE half operand bits -- a labelled extra leg,
                          * not the headline precision (BASELINE config 3 names bf16); same workspace sizes as SO3X_PREC_BF16 */

#define SO3X_KNOTS 1000      /* CDF knots, distributions.py:15                    */
#define SO3X_TRAP 999        /* CDF row length, distributions.py:26-30            */
#define SO3X_MLP_PARAMS 17358 /* RotPredict(d_model=65, skewvec), so3_train.py:26-36 */
#define SO3X_MLP_PARAMS_ROTMAT 17556 /* out_type="rotmat": Linear(65, 6) head */
#define SO3X_SCHED_ROWS 13

typedef void* so3x_stream_t; /* hipStream_t */

/* ------------------------------------------------------------------ host-side */
int so3x_abi_version(void);
const char* so3x_error_string(int code);

/* GaussianDiffusion.__init__ buffers (diffusion.py:62-92) from float64 betas.
 * [host] betas[T] -> [host] out[13][T] fp32, rows:
 *  0 betas 1 alphas_cumprod 2 alphas_cumprod_prev 3 sqrt_alphas_cumprod
 *  4 sqrt_one_minus_alphas_cumprod 5 log_one_minus_alphas_cumprod
 *  6 sqrt_recip_alphas_cumprod 7 sqrt_recipm1_alphas_cumprod 8 posterior_variance
 *  9 posterior_log_variance_clipped 10 posterior_mean_coef1 11 posterior_mean_coef2
 *  12 posterior sigma = exp(0.5*row 9)  (diffusion.py:324) */
int so3x_schedule_from_betas(const double* betas, int T, float* out);

/* cosine_beta_schedule(T, s=0.008) of the un-vendored denoising_diffusion_pytorch
 * helper (diffusion.py:60 call site).  [host] betas[T] float64. */
int so3x_cosine_beta_schedule(int T, double* betas);

/* The reference's fp32 knot vector pi*linspace(0,1,1000)**3 (distributions.py:15)
 * and Haar weights (1-cos knots)/pi evaluated in fp32 (distributions.py:21),
 * compiled in from the golden fixture.  [host] out arrays of 1000 floats. */
int so3x_igso3_knots(float* knots, float* haar_w);

/* SinusoidalPosEmb frequencies, models.py:18-21.  [host] out[half_dim]. */
int so3x_posemb_freqs(int half_dim, float* out);

/* ------------------------------------------------------------ rotation algebra */
/* util.py:222-252  quat_to_rmat: q[n][4] (real first, any norm) -> R[n][3][3] */
int so3x_quat_to_rmat(so3x_stream_t s, const float* q, float* R, int64_t n);
/* util.py:164-192  log_rmat: R -> skew-symmetric log [n][3][3] */
int so3x_log_rmat(so3x_stream_t s, const float* R, float* log_out, int64_t n);
/* util.py:164-192 + 79-84  skew2vec(log_rmat(R)): R -> [n][3] */
int so3x_log_rmat_vec(so3x_stream_t s, const float* R, float* vec_out, int64_t n);
/* util.py:95-107  orthogonalise: M[n][3][3] (any matrix) -> U round(S) V^T of its SVD (singular values snapped to integers) */
int so3x_orthogonalise(so3x_stream_t s, const float* M, float* out, int64_t n);
/* torch.matrix_exp(vec2skew(v)) (diffusion.py:294; util.py:87-92): v[n][3] -> R */
int so3x_exp_skewvec(so3x_stream_t s, const float* v, float* R, int64_t n);
/* util.py:349-361  so3_scale: exp(k log R); k_stride 0 = one scalar for all, 1 = per sample */
int so3x_so3_scale(so3x_stream_t s, const float* R, const float* k, int64_t k_stride, float* out, int64_t n);
/* util.py:195-205  aa_to_rmat: axis[n][3] (any norm), angle[n] -> R */
int so3x_aa_to_rmat(so3x_stream_t s, const float* axis, const float* angle, float* R, int64_t n);
/* util.py:208-219  rmat_to_aa: R -> axis[n][3] (NaN at angle 0, as the reference), angle[n] */
int so3x_rmat_to_aa(so3x_stream_t s, const float* R, float* axis, float* angle, int64_t n);
/* util.py:325-338  so3_lerp: a_stride 0 (one matrix for all, e.g. identity) or 9; w_stride 0 or 1 */
int so3x_so3_lerp(so3x_stream_t s, const float* a, int64_t a_stride, const float* b, const float* w,
                  int64_t w_stride, float* out, int64_t n);
/* util.py:315-322  rmat_dist: ||log(a^T b)||_F -> [n] */
int so3x_rmat_dist(so3x_stream_t s, const float* a, const float* b, float* out, int64_t n);
/* util.py:67-76  six2rmat: x6[n][6] -> R[n][3][3] (rows b1, b2, b1 x b2 of the Gram-Schmidt of the two 3-vectors), and its
 * autograd: dR = dL/dR[n][3][3] -> dx6[n][6] (the reference differentiates it with torch autograd, so3_train.py:47-48) */
int so3x_six2rmat(so3x_stream_t s, const float* x6, float* R, int64_t n);
int so3x_six2rmat_bwd(so3x_stream_t s, const float* x6, const float* dR, float* dx6, int64_t n);
/* autograd of log_rmat (util.py:164-175, generic branch) and of rmat_dist (util.py:315-322), needed by
 * loss_type="prevstep" (diffusion.py:358-365): dlog[n][3][3] -> dR;  ddist[n] -> (da, db) */
int so3x_log_rmat_bwd(so3x_stream_t s, const float* R, const float* dlog, float* dR, int64_t n);
int so3x_rmat_dist_bwd(so3x_stream_t s, const float* a, const float* b, const float* ddist, float* da, float* db, int64_t n);
/* batched a @ b and a @ b^T (diffusion.py:297,302,326,346); strides 0 (broadcast) or 9 */
int so3x_rmul(so3x_stream_t s, const float* a, int64_t a_stride, const float* b, int64_t b_stride,
              int transpose_b, float* out, int64_t n);

/* --------------------------------------------------------------------- IGSO(3) */
/* distributions.py:53-72  _eps_ft pointwise (float64 inside, fp32 out): omega[n], eps (stride 0/1) */
int so3x_igso3_eps_ft(so3x_stream_t s, const float* omega, const float* eps, int64_t eps_stride,
                      float* out, int64_t n);
/* distributions.py:15-30  CDF table rows: eps[n_rows] -> trap[n_rows][999] */
int so3x_igso3_build_tables(so3x_stream_t s, const float* eps, int64_t n_rows, float* trap);
/* Search guide for rows that are looked up per sample (not in the reference: a lookup accelerator with bit-identical
 * results): guide[n_rows][SO3X_GUIDE_PITCH] uint16, guide[r][b] = #{k : trap[r][k] <= b / SO3X_GUIDE_BINS} for
 * b = 0..SO3X_GUIDE_BINS (one pad entry keeps rows 4-byte aligned).  Every `guide` argument below is optional
 * (NULL = full bisection). */
#define SO3X_GUIDE_BINS 256
#define SO3X_GUIDE_PITCH 258
int so3x_igso3_build_guide(so3x_stream_t s, const float* trap, int64_t n_rows, uint16_t* guide);
/* distributions.py:33-51  inverse-CDF sampling.
 *  trap [n_rows][999]; row_idx int64[n] or NULL (then every sample uses row `row_const`);
 *  quirk_col0 != 0: the interpolation weight is gathered from the row of sample 0
 *      (row_idx[0]) -- distributions.py:42-43 with batched eps; 0 = own row.
 *  axes[n][3] / unif[n]: explicit draws (parity runs); both NULL = in-kernel Philox4x32-10
 *      keyed by seed, counter = (index_base + i, rng_offset) so results do not depend on
 *      launch geometry or GPU count.
 *  mean: 9 floats or NULL (identity).  angle_out[n], axis_out[n][3] optional. */
int so3x_igso3_sample(so3x_stream_t s, const float* trap, const uint16_t* guide, const int64_t* row_idx, int64_t row_const,
                      int quirk_col0, const float* axes, const float* unif, uint64_t seed,
                      uint64_t rng_offset, int64_t index_base, const float* mean, float* out,
                      float* angle_out, float* axis_out, int64_t n);
/* distributions.py:74-77 log_prob, and its gradient (distributions.py:189-190 obtains it by
 * autograd): logp[n]; score_vec[n][3] = (f'/f)*axis (tangent form) and/or grad_R[n][3][3]
 * = d logp / dR (autograd-shaped); either may be NULL. */
int so3x_igso3_logprob_score(so3x_stream_t s, const float* R, const float* eps, int64_t eps_stride,
                             float* logp, float* score_vec, float* grad_R, int64_t n);

/* ------------------------------------------------------------------- score MLP */
/* models.py:13-25 + so3_train.py:11-49.  n_out = 3 (out_type="skewvec": params = the 17,358 fp32 values in
 * state_dict order net.{0,2,4,6,8}.{weight,bias}) or 6 (out_type="rotmat", so3_train.py:21-22: net.8 is
 * Linear(65, 6), 17,556 values; the op returns the RAW 6 outputs, six2rmat is so3x_six2rmat).  t int64, t_stride
 * 0 (one timestep for the whole batch, the (1,)-shaped t of so3_test.py:31) or 1.
 * t_table: 0 = timesteps are arbitrary, the embedding is evaluated per sample in-kernel;
 *          T > 0 = the caller guarantees 0 <= t < T for every sample (SO3Diffusion's
 *          num_timesteps): the call builds [T][96] effective-bias / [T][56] embedding tables
 *          (appendix C.3 of SURVEY.md) and gathers rows by t -- no per-sample sin/cos. */
size_t so3x_mlp_workspace_bytes(int64_t n, int precision, int t_table);
int so3x_mlp_fwd(so3x_stream_t s, const float* params, const float* R, const int64_t* t,
                 int64_t t_stride, float* out, int64_t n, int n_out, int precision, int t_table,
                 void* workspace, size_t workspace_bytes);
/* Training forward (bf16 operands and t_table > 0 only, else SO3X_ERR_UNSUPPORTED): the same output plus the four
 * layers' pre-activations parked in `zstash` (so3x_mlp_stash_bytes(n) = 544 B per sample, f16) for so3x_mlp_bwd. */
size_t so3x_mlp_stash_bytes(int64_t n);
int so3x_mlp_fwd_stash(so3x_stream_t s, const float* params, const float* R, const int64_t* t,
                       int64_t t_stride, float* out, void* zstash, int64_t n, int n_out, int precision, int t_table,
                       void* workspace, size_t workspace_bytes);
/* autograd of the above for a given dL/dout[n][n_out] -> dparams[17358 | 17556] (overwritten).  zstash: NULL (the forward is
 * recomputed inside) or the stash written by so3x_mlp_fwd_stash for the SAME params, R, t (bf16, t_table > 0). */
int so3x_mlp_bwd(so3x_stream_t s, const float* params, const float* R, const int64_t* t,
                 int64_t t_stride, const float* dout, float* dparams, int64_t n, int n_out, int precision,
                 int t_table, const void* zstash, void* workspace, size_t workspace_bytes);

/* ------------------------------------------------------------- diffusion steps */
/* SO3Diffusion.q_sample + the p_losses target (diffusion.py:339-355), fused with the
 * noise draw: noise ~ IGSO3(sqrt(1-abar_t)) from trap_q rows [T][999];
 * x_t = so3_scale(x0, sqrt(abar_t)) @ noise; target = vee(log noise)/eps_t.
 * sched = device copy of the [13][T] table.  noise_in != NULL teacher-forces the noise
 * (then trap_q/axes/unif are ignored).  x_t, target, noise_out optional outputs.
 * rng_offset_dev (optional): a device-resident int64 added to rng_offset at run time, so that a captured hipGraph of a
 * training step draws fresh noise on every replay (the caller increments it inside the graph).
 * t is clamped to [0, T-1] (as so3x_p_mean_t does): an out-of-range timestep reads the nearest table row instead of
 * memory outside the tables (the reference raises IndexError, diffusion.py:16). */
int so3x_q_sample_target(so3x_stream_t s, const float* sched, int T, const float* trap_q, const uint16_t* guide_q,
                         const float* x0, const int64_t* t, int quirk_col0, const float* noise_in,
                         const float* axes, const float* unif, uint64_t seed, uint64_t rng_offset, const int64_t* rng_offset_dev,
                         int64_t index_base, float* x_t, float* target, float* noise_out, int64_t n);

/* Reverse mean for a given network output v (diffusion.py:291-313): x0hat (optional) and
 * posterior mean, one shared timestep t. */
int so3x_p_mean(so3x_stream_t s, const float* sched, int T, const float* x, const float* v, int t,
                float* x0hat, float* mean, int64_t n);
/* The same with the timestep read on the device, per sample (t_stride 1: the reference's extract(coef, t, shape) of
 * predict_start_from_noise / q_posterior, diffusion.py:291-306) or one for all (t_stride 0: no host copy of t needed);
 * values outside [0, T) are clamped. */
int so3x_p_mean_t(so3x_stream_t s, const float* sched, int T, const float* x, const float* v, const int64_t* t, int64_t t_stride,
                  float* x0hat, float* mean, int64_t n);

/* SO3Diffusion.p_sample / p_sample_loop (diffusion.py:315-337) with the RotPredict score
 * network fused in: applies n_steps reverse steps t_start, t_start-1, ... to x in place
 * (x_out may alias x_in).  trap_p = CDF rows of the posterior sigma [T][999].
 * axes/unif: explicit draws for ONE step (n_steps must be 1), else in-kernel Philox with
 * counter (index_base + i, rng_offset + t).  guide_p: optional search guide of trap_p (so3x_igso3_build_guide).
 * Diagnostics: every launch leaves two uint64 at byte so3x_p_sample_clock_offset(T, precision) of its workspace -- the
 * shader-clock ticks and the 100 MHz reference ticks that wave 0 of workgroup 0 spent in the kernel; ticks / reference
 * x 100 MHz = the clock the launch ran at (the chip lowers it under load; bench.py prices the vector issue port with it). */
size_t so3x_p_sample_workspace_bytes(int T, int precision);
size_t so3x_p_sample_clock_offset(int T, int precision);
int so3x_p_sample_chain(so3x_stream_t s, const float* params, const float* sched, int T,
                        const float* trap_p, const uint16_t* guide_p, const float* x_in, float* x_out, int t_start,
                        int n_steps, const float* axes, const float* unif, uint64_t seed,
                        uint64_t rng_offset, int64_t index_base, int64_t n, int precision,
                        void* workspace, size_t workspace_bytes);

/* The same in two calls, for callers that drive the chain ONE STEP PER CALL (so3_test.py:24-31: `R = process.p_sample(R, t)`
 * in a Python loop; diffusion.py:335-336): everything so3x_p_sample_chain derives from the parameters and the tables -- weight
 * image, per-timestep effective-bias rows and layer-0 fragments, CDF records -- depends on neither x nor t, so
 *   so3x_p_sample_prepare   builds it ONCE for all T timesteps into `workspace` (so3x_p_sample_workspace_bytes), and
 *   so3x_p_sample_prepared  runs n_steps reverse steps from that workspace: one kernel launch, no preparation.
 * The caller re-prepares whenever params, trap_p or guide_p change (so3x.diffusion.SO3Diffusion keys a cache on the parameter
 * buffer's version).  t_dev (optional, device int64[1]): the first timestep is READ ON THE DEVICE (clamped to
 * [n_steps - 1, T - 1]) instead of taken from t_start -- the caller's `t` tensor goes straight through, with no host copy and no
 * synchronisation (the reference's own loop synchronises every step, diffusion.py:320).  Other arguments as so3x_p_sample_chain. */
int so3x_p_sample_prepare(so3x_stream_t s, const float* params, int T, const float* trap_p, const uint16_t* guide_p, int precision,
                          void* workspace, size_t workspace_bytes);
int so3x_p_sample_prepared(so3x_stream_t s, const float* sched, int T, const float* trap_p, const uint16_t* guide_p, const float* x_in,
                           float* x_out, int t_start, const int64_t* t_dev, int n_steps, const float* axes, const float* unif, uint64_t seed,
                           uint64_t rng_offset, int64_t index_base, int64_t n, int precision, void* workspace, size_t workspace_bytes);

/* ------------------------------------------- wide residual score network (8f row 3) */
/* so3_lock_train.RotPredict (so3_lock_train.py:11-59): d_model = 255, input
 * [R(9), sin(123), cos(123)] (models.py:13-25 with dim 246), six ResLayer(Linear(255,255)+SiLU) blocks
 * (models.py:28-34), Linear(255, n_out).  n_out = 3 (out_type="skewvec": params = the 392,448 fp32 values in state_dict
 * order net.{0..5}.layer.0.{weight,bias}, net.6.{weight,bias}) or 6 (out_type="rotmat": 393,216 values, raw outputs).  t_table = T > 0 is REQUIRED: the call builds the
 * [T][256] input-row table and gathers by t; timesteps outside [0, T) are clamped into it. */
#define SO3X_RESNET_D 255
#define SO3X_RESNET_PARAMS 392448
#define SO3X_RESNET_PARAMS_ROTMAT 393216
size_t so3x_resnet_workspace_bytes(int precision, int t_table);
int so3x_resnet_fwd(so3x_stream_t s, const float* params, const float* R, const int64_t* t,
                    int64_t t_stride, float* out, int64_t n, int n_out, int precision, int t_table,
                    void* workspace, size_t workspace_bytes);
/* autograd of so3x_resnet_fwd for a given dL/dout[n][n_out] -> dparams[392448 | 393216] (overwritten): forward recomputed with
 * its layer inputs and pre-activations parked in the workspace (~10 KB per sample with bf16 operands, ~20 KB in
 * fp32), dX chain and dW GEMMs on the matrix cores, deterministic reduction. */
size_t so3x_resnet_train_workspace_bytes(int64_t n, int precision, int t_table);
/* Training forward: the output plus the layer inputs / pre-activations in `stash` (so3x_resnet_stash_bytes: 6.5 KB per
 * sample with bf16 operands, 13 KB in fp32) for so3x_resnet_bwd; workspace = so3x_resnet_workspace_bytes. */
size_t so3x_resnet_stash_bytes(int64_t n, int precision);
int so3x_resnet_fwd_stash(so3x_stream_t s, const float* params, const float* R, const int64_t* t,
                          int64_t t_stride, float* out, void* stash, int64_t n, int n_out, int precision, int t_table,
                          void* workspace, size_t workspace_bytes);
/* stash: NULL (the forward is run again inside) or what so3x_resnet_fwd_stash wrote for the SAME params, R, t, precision */
int so3x_resnet_bwd(so3x_stream_t s, const float* params, const float* R, const int64_t* t,
                    int64_t t_stride, const float* dout, float* dparams, int64_t n, int n_out, int precision,
                    int t_table, const void* stash, void* workspace, size_t workspace_bytes);
/* so3x_p_sample_chain with this network as the denoiser (so3_lock_test.py:24-31); workspace =
 * so3x_resnet_workspace_bytes(precision, T). */
int so3x_resnet_p_sample_chain(so3x_stream_t s, const float* params, const float* sched, int T,
                               const float* trap_p, const uint16_t* guide_p, const float* x_in, float* x_out, int t_start,
                               int n_steps, const float* axes, const float* unif, uint64_t seed,
                               uint64_t rng_offset, int64_t index_base, int64_t n, int precision,
                               void* workspace, size_t workspace_bytes);
/* The same in two calls, as so3x_p_sample_prepare / so3x_p_sample_prepared for the 65-wide network (so3_lock_test.py:24-31 drives
 * the chain one `p_sample` per call): the weight image and the [T][256] input-row table are built once per set of parameters
 * (workspace = so3x_resnet_workspace_bytes(precision, T)); t_dev (optional, device int64[1]): the first timestep read on the
 * device, clamped to [n_steps - 1, T - 1]. */
int so3x_resnet_p_sample_prepare(so3x_stream_t s, const float* params, int T, int precision, void* workspace, size_t workspace_bytes);
int so3x_resnet_p_sample_prepared(so3x_stream_t s, const float* sched, int T, const float* trap_p, const uint16_t* guide_p, const float* x_in,
                                  float* x_out, int t_start, const int64_t* t_dev, int n_steps, const float* axes, const float* unif, uint64_t seed,
                                  uint64_t rng_offset, int64_t index_base, int64_t n, int precision, const void* workspace,
                                  size_t workspace_bytes);

/* ------------------------------------------------------ SE(3) = SO(3) x R^3 layer */
/* SE3Diffusion.q_sample + the two p_losses targets (diffusion.py:496-513) fused with the
 * IGSO3xR3 noise draw (distributions.py:84-101): rotation noise as so3x_q_sample_target,
 * shift noise = z * (eps_t * shift_scale), z ~ N(0,1)^3;
 *   xt_rot = so3_scale(x0_rot, sqrt(abar)) @ noise_rot,  xt_shift = x0_shift*sqrt(abar) + noise_shift,
 *   target_rot = vee(log noise_rot)/eps,  target_shift = noise_shift/(eps*shift_scale).
 * axes/unif/znorm: explicit draws (all three or none; none = in-kernel Philox + Box-Muller). */
int so3x_se3_q_sample_target(so3x_stream_t s, const float* sched, int T, const float* trap_q, const uint16_t* guide_q,
                             float shift_scale, const float* x0_rot, const float* x0_shift,
                             const int64_t* t, int quirk_col0, const float* axes, const float* unif,
                             const float* znorm, uint64_t seed, uint64_t rng_offset, int64_t index_base,
                             float* xt_rot, float* xt_shift, float* target_rot, float* target_shift,
                             int64_t n);
/* SE3Diffusion.predict_start_from_noise + q_posterior (diffusion.py:444-466) for one shared t:
 * v_rot/v_shift = the denoiser's AffineGrad (rot_g [n][3], shift_g [n][3]). */
int so3x_se3_p_mean(so3x_stream_t s, const float* sched, int T, const float* x_rot, const float* x_shift,
                    const float* v_rot, const float* v_shift, int t, float* mean_rot, float* mean_shift,
                    int64_t n);
/* SE3Diffusion.p_sample noise (diffusion.py:476-483): out_rot = mean_rot @ noise, out_shift =
 * mean_shift + sigma*shift_scale*z.  trap_row = the 999-entry CDF row of sigma.  shared_rot != 0
 * reproduces the reference's ONE rotation noise for the whole batch (scalar eps + empty sample
 * shape, distributions.py:98-101): axes[3]/unif[1] are then single draws; 0 = one per sample. */
int so3x_se3_p_noise(so3x_stream_t s, const float* trap_row, float sigma, float shift_scale,
                     const float* mean_rot, const float* mean_shift, const float* axes, const float* unif,
                     const float* znorm, uint64_t seed, uint64_t rng_offset, int64_t index_base,
                     int shared_rot, float* out_rot, float* out_shift, int64_t n);
/* move_prot (prot_util.py:73-81): S rigid transforms applied to S structures of L residues:
 * pos' = (pos - mean_L(pos)) R^T + mean + shift, frames' = frames R^T (frames may be NULL). */
int so3x_rigid_move(so3x_stream_t s, const float* rot, const float* shift, const float* pos,
                    const float* frames, float* out_pos, float* out_frames, int64_t S, int64_t L);
/* the same for S structures of DIFFERENT lengths concatenated along the residue axis: structure c = rows off[c] .. off[c + 1]
 * (off int64 [S + 1]) -- what ProtProjection (prot_util.py:102-117) does to the ligands of a batch, one move_prot each, in ONE launch */
int so3x_rigid_move_ragged(so3x_stream_t s, const float* rot, const float* shift, const float* pos, const float* frames, const int64_t* off,
                           float* out_pos, float* out_frames, int64_t S);

/* PointCloudProj (models.py:75-91, so3 = True): the projection the Projected*Diffusion variants (diffusion.py:377-429,
 * 525-573) feed their denoisers with: out[n][P][3] = cloud @ rot[n]^T.  cloud_stride 0: one cloud [P][3] for every rotation;
 * 3 P: one cloud per rotation, cloud[n][P][3] (how aircraft_rotate.py:104-106 calls it: a batch of shapes, one pose each). */
int so3x_rotate_cloud(so3x_stream_t s, const float* rot, const float* cloud, int64_t cloud_stride, float* out, int64_t n, int64_t P);

/* -------------------------------------------- PlaneNet: the point-cloud pose denoiser (8f row 4)
 * models.PlaneNet (models.py:185-210), the denoise_fn ProjectedSO3Diffusion trains in aircraft_rotate.py:64-108:
 *   x [B][P][3] points, t int64 [B]  ->
 *   h0 = [ Siren(3 -> dim/2, scale 30): post_scale(sin(positional(x)))  (models.py:50-72)  ||  SinusoidalPosEmb(dim/2)(t)  (models.py:13-25) ]
 *   `layers` x nn.TransformerEncoderLayer(dim, heads): post-norm, ReLU, feed-forward `ffn` (torch's default 2048), LayerNorm eps
 *        1e-5, attention over the P points of one cloud, no mask; dropout_p = 0: eval-mode arithmetic (the validation pass,
 *        aircraft_rotate.py:113-117, and ProjectedSO3Diffusion.p_sample); dropout_p > 0: the training-mode forward the
 *        reference trains with (aircraft_rotate.py:66 `net.train()` on nn.TransformerEncoderLayer's default p = 0.1), see below
 *   PoolRN(dim) with every point unmasked (models.py:94-110)  ->  Linear(dim, 3)            -> out [B][3]
 * params: the fp32 values in state_dict order -- encoder.layers.{l}.{self_attn.in_proj_weight, self_attn.in_proj_bias,
 *   self_attn.out_proj.weight, self_attn.out_proj.bias, linear1.weight, linear1.bias, linear2.weight, linear2.bias, norm1.weight,
 *   norm1.bias, norm2.weight, norm2.bias}, position_siren.{positional.weight, positional.bias, post_scale.weight, post_scale.bias},
 *   out_net.0.pool.0.{weight, bias}, out_net.0.lin.{weight, bias}, out_net.1.{weight, bias}: so3x_planenet_param_count values.
 * precision SO3X_PREC_F32: every product on the exact-fp32 MFMA, any (dim, heads, layers, ffn) with dim % heads == 0, dim % 4 == 0;
 *           SO3X_PREC_BF16: bf16 operands / fp32 accumulate, activations kept in bf16; dim = 512, heads = 4 (head width 128),
 *           ffn = 2048, P % 64 == 0, layers <= 4 only (the aircraft task's shape), else SO3X_ERR_UNSUPPORTED -- from the
 *           forward already, so that no stash is built for a backward that would refuse it.
 * so3x_planenet_fwd: stash == NULL: inference (layer buffers reused inside the workspace).  stash != NULL
 *   (so3x_planenet_stash_bytes): every layer's activations are kept there for so3x_planenet_bwd.  encoding_out (optional):
 *   the encoder's output [B][P][dim] fp32 (parity tests).
 * so3x_planenet_bwd: dparams[param_count] (overwritten) = d sum(out * dout) / d params for dout [B][3], from the stash the
 *   forward wrote for the SAME params, x, t.  (The inputs carry no gradient: x is a projection of the noised pose,
 *   diffusion.py:389-392.)  Deterministic: fixed-order reductions, no atomics.
 * workspace: so3x_planenet_workspace_bytes covers either call.
 * prepared_weights (optional, so3x_planenet_fwd): the bf16 form converts the weight matrices to a bf16 image on every call (25 MB,
 *   ~15 us); a caller whose parameters did not change since (sampling: hundreds of calls per set of weights) builds the image once
 *   with so3x_planenet_prepare (so3x_planenet_weights_bytes; 0 bytes / a no-op for the exact-fp32 form) and passes it here.
 *   NULL = convert inside the call.
 * dropout_p in [0, 1), seed, rng_offset: with dropout_p > 0 (needs a stash: it is a training forward) the four dropout sites of
 *   every encoder layer are active as in torch -- attention probabilities after the softmax, the attention block's output,
 *   the feed-forward's hidden activations after the ReLU, the feed-forward's output; kept values scaled by 1 / (1 - p).  The
 *   mask of flat element e of site s in layer l is 16-bit piece (e & 7) (words x, y, z, w; low half first) of
 *   Philox4x32-10(key = seed, counter = (e >> 3, rng_offset << 8 | 4 l + s)) >= floor(p 2^16) (e: row-major index into
 *   [B][heads][P][P] / [B P][dim] / [B P][ffn] / [B P][dim]):
 *   a function of (seed, rng_offset) only, so so3x_planenet_bwd -- given the SAME three values -- regenerates it.  torch's own
 *   mask stream depends on its kernels' launch geometry and cannot be matched; parity is against the reference's modules run
 *   with these masks.  Both precisions draw the same masks (the bf16 form applies them in its GEMM epilogues and, through keep
 *   bits one kernel writes into the stash per layer, inside its attention kernels). */
int64_t so3x_planenet_param_count(int dim, int heads, int layers, int ffn);
size_t so3x_planenet_workspace_bytes(int64_t B, int64_t P, int dim, int heads, int layers, int ffn, int precision);
size_t so3x_planenet_stash_bytes(int64_t B, int64_t P, int dim, int heads, int layers, int ffn, int precision);
size_t so3x_planenet_weights_bytes(int dim, int heads, int layers, int ffn, int precision);
int so3x_planenet_prepare(so3x_stream_t s, const float* params, int dim, int heads, int layers, int ffn, int precision, void* weights,
                          size_t weights_bytes);
int so3x_planenet_fwd(so3x_stream_t s, const float* params, const float* x, const int64_t* t, float* out, float* encoding_out, int64_t B,
                      int64_t P, int dim, int heads, int layers, int ffn, int precision, void* stash, void* workspace, size_t workspace_bytes,
                      const void* prepared_weights, float dropout_p, uint64_t seed, uint64_t rng_offset);
int so3x_planenet_bwd(so3x_stream_t s, const float* params, const float* x, const int64_t* t, const float* dout, float* dparams, int64_t B,
                      int64_t P, int dim, int heads, int layers, int ffn, int precision, const void* stash, void* workspace,
                      size_t workspace_bytes, float dropout_p, uint64_t seed, uint64_t rng_offset);

/* -------------------------------------------- ProtNet: the docking denoiser of prot_train.py (8f row 4)
 * models.ProtNet (models.py:212-319), the denoise_fn ProjectedSE3Diffusion trains in prot_train.py:90-108 (BASELINE config 5):
 *   a batch of B complexes = B receptor chains + B ligand chains of ANY lengths, handed over as the reference's ProtData fields
 *   concatenated along the residue axis (CSR): *_res [n][21] one-hot residue types (prot_util.py:9-35), *_pos [n][3] CA
 *   positions, *_ang [n][3][3] residue frames, *_off int64 [B + 1] (chain c = rows off[c] .. off[c + 1]); t int64 [B].
 *   per chain: [ res_conv(residues) | Siren(3 -> dim/2, scale 0.1)(positions) | Siren(9 -> dim/4)(frames) ]  (models.py:219-252, 276-286)
 *     -> rec_tf: t_depth x nn.TransformerEncoderLayer(dim, heads) [post-norm, ReLU, feed-forward 2048, eval-mode arithmetic] + the
 *        encoder's final LayerNorm, attention over the chain's own residues (src_key_padding_mask: padded keys take no mass;
 *        BOTH chain kinds go through rec_tf, models.py:288 and 302 -- lig_tf is never run and its gradient is zero)
 *     -> PoolRN(dim) and PoolPos(dim) over the chain's residues with the chain kind's pool parameters (models.py:94-127)
 *   out [B][6] = last([SinusoidalPosEmb(dim)(t) | rec pool | rec pos | lig pool | lig pos]) = (rot_g, shift_g) (models.py:261-270, 311-318)
 * params: the fp32 values in state_dict order (so3x_protnet_param_count values; the order is spelled out in csrc/so3x_protnet.hpp).
 * max_len >= every chain's length (the kernels pad to it internally); n_rec / n_lig = rows of the receptor / ligand arrays.
 * precision SO3X_PREC_F32: every product on the exact-fp32 MFMA; any dim % 4 == 0, dim % heads == 0, dim <= 1024, c_depth >= 2
 *           (prot_train.py's own defaults -- dim 1024, 8 heads, t_depth 12, c_depth 8 -- included);
 *           SO3X_PREC_BF16: the class defaults' width (dim 64, 4 heads; any t_depth <= 8, c_depth <= 8, max_len <= 256): bf16
 *           operands / fp32 accumulate, one persistent workgroup per chain with the chain's activations in LDS; forward only
 *           (stash must be NULL), else SO3X_ERR_UNSUPPORTED.
 * so3x_protnet_fwd: stash == NULL: inference.  stash != NULL (so3x_protnet_stash_bytes): activations kept for so3x_protnet_bwd.
 *   pool_out (optional) [B][3 dim + 6]: the head's input; enc_out (optional) [2 B][max_len][dim]: rec_tf's output in the padded
 *   layout (receptors first; rows past a chain's length hold what the reference's padded rows hold -- exact form -- or zeros -- bf16 form).
 * so3x_protnet_bwd: dparams[param_count] (overwritten) = d sum(out * dout) / d params for dout [B][6], from the stash of the
 *   forward.  (The inputs carry no gradient: they are a projection of the noised pose, diffusion.py:558-559.)  Deterministic.
 * dropout_p in [0, 1), seed, rng_offset (exact-fp32 form, needs a stash: it is a training forward): the training-mode arithmetic
 *   the reference trains with (prot_train.py:75 `net.train()` on nn.TransformerEncoderLayer's default p = 0.1) -- rec_tf's four
 *   dropout sites per layer, masks as so3x_planenet_fwd's (Philox4x32-10 pieces keyed by (seed, rng_offset << 8 | 4 l + site)) with
 *   the element index e taken row-major in THIS library's padded tensors: [2 B][heads][max_len][max_len] for the probabilities,
 *   [2 B max_len][dim] / [2 B max_len][2048] / [2 B max_len][dim] for the others (receptor chains first).  so3x_protnet_bwd regenerates
 *   them from the same three values.  (The bf16 form is the inference form: dropout_p must be 0.) */
int64_t so3x_protnet_param_count(int dim, int heads, int t_depth, int c_depth);
size_t so3x_protnet_workspace_bytes(int64_t B, int64_t max_len, int64_t n_rec, int64_t n_lig, int dim, int heads, int t_depth, int c_depth,
                                    int precision);
size_t so3x_protnet_stash_bytes(int64_t B, int64_t max_len, int dim, int heads, int t_depth, int c_depth, int precision);
int so3x_protnet_fwd(so3x_stream_t s, const float* params, const float* rec_res, const float* rec_pos, const float* rec_ang, const int64_t* rec_off,
                     int64_t n_rec, const float* lig_res, const float* lig_pos, const float* lig_ang, const int64_t* lig_off, int64_t n_lig,
                     const int64_t* t, float* out, float* pool_out, float* enc_out, int64_t B, int64_t max_len, int dim, int heads, int t_depth,
                     int c_depth, int precision, void* stash, void* workspace, size_t workspace_bytes, float dropout_p, uint64_t seed,
                     uint64_t rng_offset);
int so3x_protnet_bwd(so3x_stream_t s, const float* params, const float* dout, float* dparams, int64_t B, int64_t max_len, int dim, int heads,
                     int t_depth, int c_depth, int precision, const void* stash, void* workspace, size_t workspace_bytes, float dropout_p,
                     uint64_t seed, uint64_t rng_offset);

/* ------------------------------------------------------- sample-quality statistics */
/* The pair sums behind util.MMD / Ker_2samp_test (util.py:254-312):
 *   out[0] = scale * sum_{i < nx, j < ny} k(X_i, Y_j),
 * kind 0: rmat_gaussian_kernel = exp(-rmat_dist)  (util.py:128-134);
 * kind 1: rmat_cosine_kernel = (tr(Y_j^T X_i) - 1)/2  (util.py:136-151).
 * Deterministic (per-block double partial sums, fixed-order final sum). */
size_t so3x_kernel_sum_workspace_bytes(int64_t nx, int64_t ny);
int so3x_kernel_sum(so3x_stream_t s, const float* X, int64_t nx, const float* Y, int64_t ny, int kind,
                    float scale, float* out, void* workspace, size_t workspace_bytes);

/* F.mse_loss of p_losses (diffusion.py:357): loss[0] = mean((a - b)^2) over n elements
 * (deterministic two-stage reduction), and its gradient grad_a = (a - b) * 2/n * gscale[0]
 * (gscale = device-resident upstream gradient of the scalar loss, NULL = 1). */
size_t so3x_mse_workspace_bytes(int64_t n);
int so3x_mse_loss(so3x_stream_t s, const float* a, const float* b, int64_t n, float* loss, void* workspace,
                  size_t workspace_bytes);
int so3x_mse_grad(so3x_stream_t s, const float* a, const float* b, int64_t n, const float* gscale, float* grad_a);


/* loss_type="prevstep" of SO3Diffusion.p_losses (diffusion.py:358-365), fused: step = x_noisy^T q_posterior_mean(x_start,
 * x_noisy, t) (299-302), loss[0] = mean_i rmat_dist(x_recon_i, step_i)^2, and (dx_recon != NULL) its gradient
 * d loss / d x_recon [n][3][3].  step_out (optional) receives the step rotations.  sched = device [13][T] table. */
size_t so3x_prevstep_workspace_bytes(int64_t n);
int so3x_prevstep_loss(so3x_stream_t s, const float* sched, int T, const float* x_recon, const float* x_start,
                       const float* x_noisy, const int64_t* t, int64_t t_stride, int64_t n, float* loss, float* dx_recon,
                       float* step_out, void* workspace, size_t workspace_bytes);
/* The same with x_recon = six2rmat(out6) applied inside (RotPredict(out_type="rotmat") under loss_type="prevstep"):
 * out6[n][6] = the network's raw outputs, dout6 (optional) = d loss / d out6 -- six2rmat, the loss and six2rmat's backward
 * in one pass instead of three kernels. */
int so3x_prevstep_loss6(so3x_stream_t s, const float* sched, int T, const float* out6, const float* x_start,
                        const float* x_noisy, const int64_t* t, int64_t t_stride, int64_t n, float* loss, float* dout6,
                        void* workspace, size_t workspace_bytes);

/* ------------------------------------------------------------------ one training step
 * The device work of one iteration of the reference's training loop (so3_train.py:73-76: loss = process(truepos);
 * loss.backward(); optim.step()) for SO3Diffusion(RotPredict(d_model=65, out_type="skewvec"), loss_type="skewvec") with
 * bf16 MLP operands, as three calls so that a data-parallel caller can put its gradient all-reduce between the second
 * and the third.  Five kernel launches in all (prep, noising, forward + loss, backward, reduce) + the optimizer's one.
 *
 * so3x_train_fwd   SO3Diffusion.p_losses (diffusion.py:348-357): noise ~ IGSO3(sqrt(1 - abar_t)) (axes/unif: explicit
 *                  draws, else in-kernel Philox with counter (index_base + i, rng_offset + *rng_counter)), x_t =
 *                  q_sample(x0, t, noise), target = vee(log noise)/eps_t, out = RotPredict(x_t, t),
 *                  loss[0] = mean((out - target)^2), dout = d loss / d out.  zstash (so3x_mlp_stash_bytes(n)) and the
 *                  workspace carry the forward's pre-activations, weight images and tables over to so3x_train_bwd:
 *                  pass the SAME workspace, untouched, and do not change params in between.  rng_counter (optional,
 *                  device int64): read as an addend of rng_offset and incremented by one by this call, so that a
 *                  captured hipGraph of the step draws fresh noise on every replay.  out (optional): the network output.
 *                  t (optional): the timesteps, int64 [n] (p_losses(x, t)), clamped to [0, T-1] (the reference raises
 *                  IndexError outside that range; a kernel must not read outside its tables).  t == NULL: the timesteps are
 *                  drawn in the kernel (SO3Diffusion.forward's randint(0, T, (b,)), diffusion.py:373), t_i = floor(T w_i /
 *                  2^32) with w_i the spare fourth word of sample i's Philox block, so that they are, like the noise, a
 *                  function of (seed, global sample index, offset) alone.  t_used: int64 [n] OUTPUT, always -- the timesteps
 *                  the step ran with (drawn, or the caller's clamped); pass it to so3x_train_bwd as its t.  It never
 *                  aliases t.  quirk_col0 (distributions.py:42-43, "column 0 is sample 0's row"): with drawn timesteps the
 *                  row is GLOBAL sample 0's (Philox index 0 whatever index_base is: every shard of a data-parallel run uses
 *                  the single-process run's row), with given timesteps this call's t[0].
 * so3x_train_bwd   autograd of the above wrt the 17,358 parameters (so3_train.py:75): grad[17358] = gscale[0] *
 *                  d loss / d params (gscale: device-resident upstream gradient of the scalar loss, NULL = 1).
 * The same step in stages (version 5), for callers that pipeline it -- so3x.graphs.TrainStepGraph runs the noising of batch
 * k+1 on a second stream beside [slab reduction -> gradient all-reduce -> Adam] of batch k:
 *   so3x_train_fwd = so3x_train_noise (x0, draws -> x_t, t_used, the regression target inside the workspace; reads
 *                    rng_counter, does not advance it) + so3x_train_net (prep of the weight images from params + network
 *                    forward + stash + MSE and its gradient; advances rng_counter if given);
 *   so3x_train_bwd = so3x_train_bwd_partial (fused backward -> per-workgroup partial slabs inside the workspace) +
 *                    so3x_train_bwd_reduce (fixed-order sum of the slabs x gscale -> grad).
 *   Same workspace for all four; params must not change between so3x_train_net and so3x_train_bwd_partial.
 * so3x_adam_step   torch.optim.Adam.step() (so3_train.py:64,76; amsgrad = maximize = False) on flat buffers of n
 *                  floats: the gradient is multiplied by grad_scale first (1/world_size after a summed all-reduce).
 *                  step: TWO device floats, zero-initialised once by the caller: [0] = the step count, advanced by this
 *                  call (torch's state['step']), [1] = scratch.  A gradient whose FIRST entry is not finite (the all-NaN
 *                  gradient so3x_train_bwd_reduce writes after a timed-out hand-shake) makes the call a no-op: parameters, moments
 *                  and step[0] stay as they were (torch.optim.Adam would write the NaN through) -- a caller counts its calls against
 *                  step[0] to see skipped updates (so3x.optim.Adam.skipped_steps). */
size_t so3x_train_workspace_bytes(int64_t n, int T);
int so3x_train_fwd(so3x_stream_t s, const float* params, const float* sched, int T, const float* trap_q, const uint16_t* guide_q,
                   const float* x0, const int64_t* t, int64_t* t_used, int quirk_col0, const float* axes, const float* unif,
                   uint64_t seed, uint64_t rng_offset, int64_t* rng_counter, int64_t index_base, int64_t n, float* x_t, float* dout,
                   void* zstash, float* loss, float* out, void* workspace, size_t workspace_bytes);
int so3x_train_noise(so3x_stream_t s, const float* sched, int T, const float* trap_q, const uint16_t* guide_q, const float* x0,
                     const int64_t* t, int64_t* t_used, int quirk_col0, const float* axes, const float* unif, uint64_t seed,
                     uint64_t rng_offset, const int64_t* rng_counter, int64_t index_base, int64_t n, float* x_t, void* workspace,
                     size_t workspace_bytes);
int so3x_train_net(so3x_stream_t s, const float* params, int T, const float* x_t, const int64_t* t_used, int64_t n, float* dout,
                   void* zstash, float* loss, float* out, int64_t* rng_counter, void* workspace, size_t workspace_bytes);
int so3x_train_bwd_partial(so3x_stream_t s, const float* x_t, const int64_t* t, const float* dout, const void* zstash, int64_t n, int T,
                           void* workspace, size_t workspace_bytes);
int so3x_train_bwd_reduce(so3x_stream_t s, int64_t n, int T, const float* gscale, float* grad, const void* workspace,
                          size_t workspace_bytes);
/* so3x_train_bwd_reduce followed by so3x_adam_step on the same 17,358 parameters, as ONE launch (single-process training: nothing
 * sits between the reduction and the optimizer; a data-parallel step puts its all-reduce there and uses the two calls).  grad is
 * still written.  Same arithmetic, term for term: bit-identical parameters. */
int so3x_train_bwd_reduce_adam(so3x_stream_t s, int64_t n, int T, const float* gscale, float* grad, const void* workspace,
                               size_t workspace_bytes, float* params, float* exp_avg, float* exp_avg_sq, float* step, float lr, float beta1,
                               float beta2, float eps, float weight_decay, float grad_scale);
int so3x_train_bwd(so3x_stream_t s, const float* x_t, const int64_t* t, const float* dout, const void* zstash, int64_t n, int T,
                   const float* gscale, float* grad, void* workspace, size_t workspace_bytes);
int so3x_adam_step(so3x_stream_t s, float* params, const float* grad, float* exp_avg, float* exp_avg_sq, float* step, int64_t n,
                   float lr, float beta1, float beta2, float eps, float weight_decay, float grad_scale);
/* so3x_train_fused (version 6): so3x_train_fwd + so3x_train_bwd_partial as ONE kernel (after the prep launch) -- `loss =
 * process(truepos); loss.backward()` of so3_train.py:73-75 up to the per-workgroup partial dW slabs.  Noise draw, q_sample and
 * target (diffusion.py:339-355), network forward (so3_train.py:39-49), MSE and d loss / d out (diffusion.py:357), the dZ chain and
 * the dW products run per 32-sample tile inside one workgroup; x_t, target, timesteps, dout and the pre-activations never go
 * through HBM (a sample costs its 36 bytes of x0).  Same arguments and meaning as so3x_train_fwd, same Philox draws and timesteps bit for bit; x_t and the targets agree to fp32 rounding (fused multiply-adds contract differently in the two translation units), the network output to bf16-operand accuracy (Philox
 * keyed by (seed, index_base + i, rng_offset + *rng_counter); t == NULL: drawn in the kernel; quirk_col0 as there); rng_counter
 * is advanced by one when the call drew from it.  loss[0] = mean((out - target)^2).  The slabs land where
 * so3x_train_bwd_reduce / so3x_train_bwd_reduce_adam (same n, T, workspace) expect them: call one of those next for grad[17358].
 * Optional outputs (NULL = not written; parity tests and debugging): t_used int64 [n], x_t [n][3][3], out [n][3]. */
int so3x_train_fused(so3x_stream_t s, const float* params, const float* sched, int T, const float* trap_q, const uint16_t* guide_q,
                     const float* x0, const int64_t* t, int64_t* t_used, int quirk_col0, const float* axes, const float* unif,
                     uint64_t seed, uint64_t rng_offset, int64_t* rng_counter, int64_t index_base, int64_t n, float* loss, float* x_t,
                     float* out, void* workspace, size_t workspace_bytes);

#ifdef __cplusplus
}
#endif
#endif /* SO3X_H */
