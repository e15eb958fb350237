/*
 * so3_oracle.c -- CPU oracle for the SO(3) diffusion hot path.
 *
 * *** TEST INFRASTRUCTURE, NOT PRODUCT CODE. ***
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.  The product path (diffusion-extensions_amd/) never
 * links, imports or falls back to it: it fails loudly without libso3x.so.
 *
 * What it is: a plain-C restatement of the reference's algorithm for the path
 * (reference = /root/reference, pure Python/PyTorch; citations are file:line
 * in that tree).  The reference's fp32-with-fp64-islands arithmetic is kept:
 * _eps_ft in double, tables/angles in fp32.  Rotation math is built twice,
 * REAL=float ("_f32": the reference's working precision) and REAL=double
 * ("_f64": the truth the conditioning-aware gates compare against).
 *
 * Parity pinning: the reference has no tests of its own;
 * this oracle is pinned against golden vectors produced by importing the
 * reference in the build container (tools/make_golden.py -> tests/golden/ npz files,
 * checked by tests/test_oracle_golden.py).  torch.matrix_exp / torch.svd are
 * replaced by closed forms, so those pins are numerical gates (G1/G2), and the
 * un-vendored denoising_diffusion_pytorch schedule helper is restated from its
 * published definition: the cosine schedule is "parity unpinned"; betas travel
 * as an explicit fixture.
 *
 * Third-party arithmetic absent from /root/reference: module
 * denoising-diffusion-pytorch (fork github.com/qazwsxal/denoising-diffusion-pytorch,
 * no pinned version, empty submodule dir; call sites diffusion.py:8-14,60).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* ------------------------------------------------------------------------ */
/* precision-generic part, built twice                                        */
/* ------------------------------------------------------------------------ */
#define REAL float
#define SUFFIX _f32
#define SQRT sqrtf
#define SIN sinf
#define COS cosf
#define ATAN2 atan2f
#define EXP expf
#include "so3_oracle_impl.h"
#undef REAL
#undef SUFFIX
#undef SQRT
#undef SIN
#undef COS
#undef ATAN2
#undef EXP

#define REAL double
#define SUFFIX _f64
#define SQRT sqrt
#define SIN sin
#define COS cos
#define ATAN2 atan2
#define EXP exp
#include "so3_oracle_impl.h"
#undef REAL
#undef SUFFIX
#undef SQRT
#undef SIN
#undef COS
#undef ATAN2
#undef EXP

/* ------------------------------------------------------------------------ */
/* precision-independent part                                                 */
/* ------------------------------------------------------------------------ */

/* cosine_beta_schedule(T, s=0.008): restated from the published lucidrains
 * denoising-diffusion-pytorch definition (un-vendored; diffusion.py:60 call site).
 * numpy float64: x = linspace(0, T+1, T+1); ac = cos(((x/(T+1))+s)/(1+s)*pi/2)^2;
 * ac /= ac[0]; betas = clip(1 - ac[1:]/ac[:-1], 0, 0.999).   PARITY UNPINNED. */
void so3o_cosine_beta_schedule(int T, double* betas) {
  int steps = T + 1;
  double s = 0.008;
  double* ac = (double*)malloc(sizeof(double) * steps);
  for (int i = 0; i < steps; i++) {
    /* np.linspace(0, steps, steps): step = steps/(steps-1) */
    double x = (i == steps - 1) ? (double)steps : (double)i * ((double)steps / (double)(steps - 1));
    double c = cos(((x / steps) + s) / (1 + s) * M_PI * 0.5);
    ac[i] = c * c;
  }
  double a0 = ac[0];
  for (int i = 0; i < steps; i++) ac[i] /= a0;
  for (int i = 0; i < T; i++) {
    double b = 1 - ac[i + 1] / ac[i];
    betas[i] = b < 0 ? 0 : (b > 0.999 ? 0.999 : b);
  }
  free(ac);
}

/* GaussianDiffusion.__init__ buffers: diffusion.py:62-92.  Computed in float64,
 * stored fp32.  Row order of out[12][T]:
 *  0 betas, 1 alphas_cumprod, 2 alphas_cumprod_prev, 3 sqrt_alphas_cumprod,
 *  4 sqrt_one_minus_alphas_cumprod, 5 log_one_minus_alphas_cumprod,
 *  6 sqrt_recip_alphas_cumprod, 7 sqrt_recipm1_alphas_cumprod, 8 posterior_variance,
 *  9 posterior_log_variance_clipped, 10 posterior_mean_coef1, 11 posterior_mean_coef2 */
void so3o_schedule_from_betas(const double* betas, int T, float* out) {
  double ac = 1.0, ac_prev;
  for (int i = 0; i < T; i++) {
    double alpha = 1.0 - betas[i];
    ac_prev = (i == 0) ? 1.0 : ac;
    ac = (i == 0) ? alpha : ac * alpha; /* np.cumprod */
    double pv = betas[i] * (1. - ac_prev) / (1. - ac);
    out[0 * T + i] = (float)betas[i];
    out[1 * T + i] = (float)ac;
    out[2 * T + i] = (float)ac_prev;
    out[3 * T + i] = (float)sqrt(ac);
    out[4 * T + i] = (float)sqrt(1. - ac);
    out[5 * T + i] = (float)log(1. - ac);
    out[6 * T + i] = (float)sqrt(1. / ac);
    out[7 * T + i] = (float)sqrt(1. / ac - 1);
    out[8 * T + i] = (float)pv;
    out[9 * T + i] = (float)log(pv > 1e-20 ? pv : 1e-20);
    out[10 * T + i] = (float)(betas[i] * sqrt(ac_prev) / (1. - ac));
    out[11 * T + i] = (float)((1. - ac_prev) * sqrt(alpha) / (1. - ac));
  }
}

/* SinusoidalPosEmb frequencies: models.py:18-21, exp(arange(half) * -(ln 1e4/(half-1))) in fp32 */
void so3o_posemb_freqs(int half_dim, float* out) {
  double emb = log(10000.0) / (half_dim - 1);
  for (int i = 0; i < half_dim; i++) out[i] = expf((float)i * (float)(-emb));
}

/* IsotropicGaussianSO3._eps_ft: distributions.py:53-72, all float64.
 * raw = 1: value before the .float() cast (double out) */
static double eps_ft_d(double t, double eps_f32_as_double) {
  const double pi = M_PI;
  double v = eps_f32_as_double * eps_f32_as_double; /* eps.double()**2 */
  double vals;
  if (t == 0.0) {
    /* limit branch :68-71 (applied after the inf/nan clean-up, so not cleaned) */
    vals = sqrt(pi) * (v * exp(2 * pi * pi / v) - 2 * v * exp(pi * pi / v) + 4 * pi * pi * v * exp(pi * pi / v))
           * exp(v / 4 - (2 * pi * pi) / v) / pow(v, 2.5);
    return vals;
  }
  vals = sqrt(pi) * pow(v, -1.5) * exp(v / 4) * exp(-((t / 2) * (t / 2)) / v)
         * (t - exp((-pi * pi) / v) * ((t - 2 * pi) * exp(pi * t / v) + (t + 2 * pi) * exp(-pi * t / v)))
         / (2 * sin(t / 2));
  if (isinf(vals) || isnan(vals)) vals = 0.0; /* :61-62 */
  return vals;
}

/* pointwise: omega[n] (fp32), eps[n*eps_stride] (fp32) -> fp32 (the .float() at :72) */
void so3o_eps_ft(const float* omega, const float* eps, long eps_stride, float* out, long n) {
  for (long i = 0; i < n; i++) out[i] = (float)eps_ft_d((double)omega[i], (double)eps[i * eps_stride]);
}

/* f'(omega)/f(omega) in double: derivative of the closed form above */
void so3o_igso3_dlogf(const float* omega, const float* eps, long eps_stride, double* out, long n) {
  const double pi = M_PI;
  for (long i = 0; i < n; i++) {
    double t = omega[i], e = eps[i * eps_stride], v = e * e;
    double E = exp(-pi * pi / v), ep = exp(pi * t / v), em = exp(-pi * t / v);
    double g = t - E * ((t - 2 * pi) * ep + (t + 2 * pi) * em);
    double gp = 1 - E * (ep + (t - 2 * pi) * (pi / v) * ep + em - (t + 2 * pi) * (pi / v) * em);
    out[i] = -t / (2 * v) + gp / g - 0.5 * cos(t / 2) / sin(t / 2);
  }
}

/* IsotropicGaussianSO3.__init__ CDF table: distributions.py:15-30.
 * knots[1000] / haar_w[1000] are the reference's fp32 vectors (fixture
 * tests/golden/igso3_knots.npz): knots = pi*linspace(0,1,1000)**3 (:15),
 * haar_w = (1-cos knots)/pi in fp32 (:21).  trap[n_rows][999]. */
void so3o_igso3_build_tables(const float* eps, long n_rows, const float* knots, const float* haar_w, float* trap) {
#pragma omp parallel for schedule(static)
  for (long r = 0; r < n_rows; r++) {
    float pdf[1000];
    for (int k = 0; k < 1000; k++) {
      float f = (float)eps_ft_d((double)knots[k], (double)eps[r]);
      pdf[k] = f * haar_w[k];        /* fp32 product :21 */
      if (knots[k] == 0.f) pdf[k] = 0.f; /* :23 */
    }
    double acc = 0.0; /* torch CPU cumsum accumulates fp32 in double (acc_type) -- pinned by golden */
    float* row = trap + 999 * r;
    for (int k = 0; k < 999; k++) {
      float sum = pdf[k] + pdf[k + 1];          /* :26 */
      float dl = knots[k + 1] - knots[k];       /* :27 */
      float term = dl * sum / 2.0f;             /* :28 */
      acc += (double)term;
      row[k] = (float)acc;
    }
    float last = row[998];
    for (int k = 0; k < 999; k++) row[k] = row[k] / last; /* :29 */
  }
}

/* IsotropicGaussianSO3.log_prob: distributions.py:74-77 -- log(float(f(angle(R)))) in fp32 */
void so3o_igso3_log_prob(const float* R, const float* eps, long eps_stride, float* logp, long n) {
  for (long b = 0; b < n; b++) {
    float w[3];
    so3o_log3_f32(R + 9 * b, w);
    float ang = sqrtf(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
    float f = (float)eps_ft_d((double)ang, (double)eps[b * eps_stride]);
    logp[b] = logf(f);
  }
}

/* One reverse step p_sample (diffusion.py:315-326) over a batch, fp32, OpenMP
 * over the batch axis.  Used as the timed CPU baseline ("port") and as the
 * step/chain checker.  sched = the 12xT table above; trap_p = posterior-sigma
 * CDF rows [T][999]; axes/unif explicit draws (ignored at t == 0). */
void so3o_p_sample_step_f32(const float* params, const float* freqs, const float* sched, int T,
                            const float* trap_p, const float* knots, const float* x, int t,
                            const float* axes, const float* unif, float* out, long n) {
  float a = sched[6 * T + t], bc = sched[7 * T + t], c1 = sched[10 * T + t], c2 = sched[11 * T + t];
#pragma omp parallel for schedule(static)
  for (long b = 0; b < n; b++) {
    float v[3], mean[9];
    long tt = t;
    so3o_mlp_fwd_one_f32(params, freqs, x + 9 * b, tt, v, NULL, 3);
    so3o_p_mean_one_f32(x + 9 * b, v, a, bc, c1, c2, NULL, mean);
    if (t == 0) { memcpy(out + 9 * b, mean, sizeof(mean)); continue; }
    float smp[9];
    so3o_igso3_sample_f32(trap_p + 999 * (long)t, NULL, -1, knots, axes + 3 * b, unif + b, NULL, smp, NULL, 1);
    so3o_mul33_f32(mean, smp, out + 9 * b);
  }
}

/* torch.optim.Adam.step() as the reference's training loops call it (so3_train.py:64,76: Adam(net.parameters(), lr=3e-4),
 * defaults betas (0.9, 0.999), eps 1e-8, weight_decay 0, amsgrad False).  The arithmetic is torch's (third-party: torch 1.8
 * per requirements.txt:2; _single_tensor_adam in torch/optim/adam.py), restated: scalars in double as torch forms them in
 * Python floats, element updates in fp32.  `step` is the count BEFORE this call (torch increments first).  Pinned by
 * tests/golden/adam.npz, produced by torch.optim.Adam itself (tools/make_golden.py adam). */
void so3o_adam_step(float* p, const float* g, float* m, float* v, long n, double step, double lr, double beta1, double beta2,
                    double eps, double weight_decay, double grad_scale) {
  const double k = step + 1.0;
  const double bc1 = 1.0 - pow(beta1, k), bc2 = 1.0 - pow(beta2, k);
  const float neg_step_size = (float)(-lr / bc1), bc2_sqrt = (float)sqrt(bc2);
  const float b1 = (float)beta1, b2 = (float)beta2, e = (float)eps, wd = (float)weight_decay, gs = (float)grad_scale;
  for (long i = 0; i < n; i++) {
    float gi = g[i] * gs;
    if (wd != 0.0f) gi = gi + wd * p[i];
    m[i] = m[i] + (1.0f - b1) * (gi - m[i]);
    v[i] = v[i] * b2 + (1.0f - b2) * gi * gi;
    p[i] = p[i] + neg_step_size * (m[i] / (sqrtf(v[i]) / bc2_sqrt + e));
  }
}

int so3o_omp_threads(void) {
#ifdef _OPENMP
  extern int omp_get_max_threads(void);
  return omp_get_max_threads();
#else
  return 1;
#endif
}
