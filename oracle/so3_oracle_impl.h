/*
 * so3_oracle_impl.h -- precision-generic body of the CPU oracle.
 *
 * TEST INFRASTRUCTURE ONLY (see so3_oracle.c header).  Included twice by
 * so3_oracle.c with REAL = float (the reference's fp32 evaluation) and
 * REAL = double (the "truth" evaluation of the same fp32 inputs used by the
 * conditioning-aware gates G1/G2 of SURVEY.md section 8c).
 *
 * Every function cites the reference file:line (under /root/reference) whose
 * algorithm it restates.  torch.matrix_exp / torch.svd on the path are
 * replaced by their closed forms (Rodrigues; SVD-orthogonalise of an already
 * orthogonal matrix is the identity map), so parity with the reference is
 * numerical (gates in tests/), never bitwise -- SURVEY.md section 8c.
 */

#define CAT_(a, b) a##b
#define CAT(a, b) CAT_(a, b)
#define FN(name) CAT(CAT(so3o_, name), SUFFIX)

/* ---- small 3x3 helpers (row-major) ---- */
static inline void FN(mul33)(const REAL* a, const REAL* b, REAL* o) {
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++)
      o[3 * i + j] = a[3 * i] * b[j] + a[3 * i + 1] * b[3 + j] + a[3 * i + 2] * b[6 + j];
}
static inline void FN(mul33_bt)(const REAL* a, const REAL* b, REAL* o) { /* a @ b^T */
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++)
      o[3 * i + j] = a[3 * i] * b[3 * j] + a[3 * i + 1] * b[3 * j + 1] + a[3 * i + 2] * b[3 * j + 2];
}
static inline void FN(mul33_at)(const REAL* a, const REAL* b, REAL* o) { /* a^T @ b */
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++)
      o[3 * i + j] = a[i] * b[j] + a[3 + i] * b[3 + j] + a[6 + i] * b[6 + j];
}

/* log_rmat as a 3-vector: util.py:164-192 followed by skew2vec (util.py:79-84).
 *   S = R - R^T; v = vee(S) = (S21, -S20, S10); s = |v|/2; c = (tr R - 1)/2;
 *   angle = atan2(s, c); scale = angle/(2 s), 0 where angle == 0 (util.py:174).
 * s == 0 with c < 0 (exact pi rotation) is the reference's NaN -> eigh branch
 * (util.py:178-191), which takes an eigenvector ROW (util.py:188, a reference
 * bug); here the mathematically correct axis from diag((R+I)/2) is used and
 * parity in that branch is unpinned. */
static inline void FN(log3)(const REAL* R, REAL* w) {
  REAL v0 = R[7] - R[5], v1 = -(R[6] - R[2]), v2 = R[3] - R[1];
  REAL s = SQRT(v0 * v0 + v1 * v1 + v2 * v2) / (REAL)2;
  REAL c = (R[0] + R[4] + R[8] - (REAL)1) / (REAL)2;
  REAL ang = ATAN2(s, c);
  if (ang == (REAL)0) { w[0] = w[1] = w[2] = (REAL)0; return; }
  if (s == (REAL)0) { /* exact pi */
    REAL d0 = (R[0] + 1) / 2, d1 = (R[4] + 1) / 2, d2 = (R[8] + 1) / 2;
    REAL a[3];
    if (d0 >= d1 && d0 >= d2) { a[0] = SQRT(d0); a[1] = (R[1] + R[3]) / (4 * a[0]); a[2] = (R[2] + R[6]) / (4 * a[0]); }
    else if (d1 >= d2)        { a[1] = SQRT(d1); a[0] = (R[1] + R[3]) / (4 * a[1]); a[2] = (R[5] + R[7]) / (4 * a[1]); }
    else                      { a[2] = SQRT(d2); a[0] = (R[2] + R[6]) / (4 * a[2]); a[1] = (R[5] + R[7]) / (4 * a[2]); }
    w[0] = ang * a[0]; w[1] = ang * a[1]; w[2] = ang * a[2];
    return;
  }
  REAL scale = ang / (2 * s);
  w[0] = scale * v0; w[1] = scale * v1; w[2] = scale * v2;
}

/* exp of hat(w): closed form of torch.matrix_exp(vec2skew(w)) (util.py:204,360;
 * diffusion.py:294).  hat (util.py:87-92): S21=w0, S20=-w1, S10=w2, antisymmetric.
 * R = I + A K + B K^2,  A = sin(th)/th, B = (1-cos th)/th^2. */
static inline void FN(exp3)(const REAL* w, REAL* R) {
  REAL x = w[0], y = w[1], z = w[2];
  REAL t2 = x * x + y * y + z * z;
  REAL th = SQRT(t2);
  REAL A, B;
  if (th < (REAL)1e-4) { A = 1 - t2 / 6; B = (REAL)0.5 - t2 / 24; }
  else { A = SIN(th) / th; B = (1 - COS(th)) / t2; }
  /* K = [[0,-z,y],[z,0,-x],[-y,x,0]] ; K^2 = w w^T - t2 I */
  R[0] = 1 + B * (x * x - t2); R[1] = -A * z + B * x * y;   R[2] = A * y + B * x * z;
  R[3] = A * z + B * x * y;    R[4] = 1 + B * (y * y - t2); R[5] = -A * x + B * y * z;
  R[6] = -A * y + B * x * z;   R[7] = A * x + B * y * z;    R[8] = 1 + B * (z * z - t2);
}

/* quat_to_rmat: util.py:222-252 (real part first, any norm) */
void FN(quat_to_rmat)(const REAL* q, REAL* out, long n) {
  for (long b = 0; b < n; b++) {
    REAL r = q[4 * b], i = q[4 * b + 1], j = q[4 * b + 2], k = q[4 * b + 3];
    REAL ts = (REAL)2 / (r * r + i * i + j * j + k * k);
    REAL* o = out + 9 * b;
    o[0] = 1 - ts * (j * j + k * k); o[1] = ts * (i * j - k * r);     o[2] = ts * (i * k + j * r);
    o[3] = ts * (i * j + k * r);     o[4] = 1 - ts * (i * i + k * k); o[5] = ts * (j * k - i * r);
    o[6] = ts * (i * k - j * r);     o[7] = ts * (j * k + i * r);     o[8] = 1 - ts * (i * i + j * j);
  }
}

/* log_rmat as vee-vector [n,3] (util.py:164-192 + 79-84) */
void FN(log_rmat_vec)(const REAL* R, REAL* w, long n) {
  for (long b = 0; b < n; b++) FN(log3)(R + 9 * b, w + 3 * b);
}

/* matrix_exp(vec2skew(w)) [n,3] -> [n,3,3] (diffusion.py:294) */
void FN(exp_vec)(const REAL* w, REAL* R, long n) {
  for (long b = 0; b < n; b++) FN(exp3)(w + 3 * b, R + 9 * b);
}

/* so3_scale: util.py:349-361 -- exp(k log R); k_stride 0 = broadcast scalar */
void FN(so3_scale)(const REAL* R, const REAL* k, long k_stride, REAL* out, long n) {
  for (long b = 0; b < n; b++) {
    REAL w[3];
    FN(log3)(R + 9 * b, w);
    REAL kk = k[b * k_stride];
    w[0] *= kk; w[1] *= kk; w[2] *= kk;
    FN(exp3)(w, out + 9 * b);
  }
}

/* aa_to_rmat: util.py:195-205 -- normalise axis (201), matrix_exp (204),
 * orthogonalise = SVD round-trip (95-107), identity on an orthogonal matrix. */
void FN(aa_to_rmat)(const REAL* axis, const REAL* ang, REAL* out, long n) {
  for (long b = 0; b < n; b++) {
    const REAL* a = axis + 3 * b;
    REAL nrm = SQRT(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]);
    REAL w[3] = {a[0] / nrm * ang[b], a[1] / nrm * ang[b], a[2] / nrm * ang[b]};
    FN(exp3)(w, out + 9 * b);
  }
}

/* rmat_to_aa: util.py:208-219 -- angle = |vee(log R)|, axis = vee/angle (NaN at 0) */
void FN(rmat_to_aa)(const REAL* R, REAL* axis, REAL* ang, long n) {
  for (long b = 0; b < n; b++) {
    REAL w[3];
    FN(log3)(R + 9 * b, w);
    REAL a = SQRT(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
    ang[b] = a;
    axis[3 * b] = w[0] / a; axis[3 * b + 1] = w[1] / a; axis[3 * b + 2] = w[2] / a;
  }
}

/* so3_lerp: util.py:325-338 -- a @ aa_to_rmat(axis(a^T b), w * angle(a^T b)) */
void FN(so3_lerp)(const REAL* A, long a_stride, const REAL* Bm, const REAL* wgt, long w_stride, REAL* out, long n) {
  for (long b = 0; b < n; b++) {
    const REAL* a = A + a_stride * b;
    REAL c[9], w[3], rc[9];
    FN(mul33_at)(a, Bm + 9 * b, c);
    FN(log3)(c, w);
    REAL ang = SQRT(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
    REAL ax[3] = {w[0] / ang, w[1] / ang, w[2] / ang};               /* rmat_to_aa, NaN at 0 */
    REAL nrm = SQRT(ax[0] * ax[0] + ax[1] * ax[1] + ax[2] * ax[2]);   /* aa_to_rmat renormalises */
    REAL ia = wgt[b * w_stride] * ang;
    REAL wi[3] = {ax[0] / nrm * ia, ax[1] / nrm * ia, ax[2] / nrm * ia};
    FN(exp3)(wi, rc);
    FN(mul33)(a, rc, out + 9 * b);
  }
}

/* rmat_dist: util.py:315-322 -- Frobenius norm of log(a^T b) = sqrt(2)*angle */
void FN(rmat_dist)(const REAL* A, const REAL* Bm, REAL* out, long n) {
  for (long b = 0; b < n; b++) {
    REAL c[9], w[3];
    FN(mul33_at)(A + 9 * b, Bm + 9 * b, c);
    FN(log3)(c, w);
    out[b] = SQRT(2 * (w[0] * w[0] + w[1] * w[1] + w[2] * w[2]));
  }
}

/* IsotropicGaussianSO3.sample: distributions.py:33-51, explicit draws.
 *   trap      [n_rows][999] fp32 CDF rows (always fp32, as the reference's)
 *   row_idx   per-sample row (NULL = row 0 for everyone: the scalar-eps branch)
 *   weight_row >= 0: row used for the trap_start/trap_end gathers -- the
 *              batched-eps "column 0" behaviour of distributions.py:42-43;
 *              -1 = own row.
 *   axes [n,3] randn draws (35), unif [n] rand draws (38), mean [9] or NULL.
 * angle interpolation runs in fp32 in both builds (the reference's dtype);
 * only the rotation construction is REAL. */
void FN(igso3_sample)(const float* trap, const long* row_idx, long weight_row, const float* knots,
                      const float* axes, const float* unif, const REAL* mean, REAL* out,
                      float* angle_out, long n) {
  for (long b = 0; b < n; b++) {
    const float* row = trap + 999 * (row_idx ? row_idx[b] : 0);
    const float* wrow = weight_row >= 0 ? trap + 999 * weight_row : row;
    float u = unif[b];
    int idx1 = 0;
    for (int k = 0; k < 999; k++) idx1 += (row[k] <= u);          /* :39 */
    if (idx1 > 998) idx1 = 998; /* unreachable for u<1 (trap[998]==1); guards the gather */
    int idx0 = idx1 - 1 < 0 ? 0 : idx1 - 1;                        /* :40 */
    float ts = wrow[idx0], te = wrow[idx1];                        /* :42-43 */
    float df = te - ts; if (df < 1e-6f) df = 1e-6f;                /* :45 */
    float wt = (u - ts) / df; wt = wt < 0.f ? 0.f : (wt > 1.f ? 1.f : wt);
    float a0 = knots[idx0 + 1], a1 = knots[idx1 + 1];              /* trap_loc = knots[1:] (:30) */
    /* torch.lerp(start,end,w): w<0.5 ? s + w*(e-s) : e - (e-s)*(1-w)  (ATen Lerp.h) */
    float dl = a1 - a0;
    float ang = wt < 0.5f ? a0 + wt * dl : a1 - dl * (1.f - wt);
    if (angle_out) angle_out[b] = ang;
    REAL ax[3] = {(REAL)axes[3 * b], (REAL)axes[3 * b + 1], (REAL)axes[3 * b + 2]};
    REAL nrm = SQRT(ax[0] * ax[0] + ax[1] * ax[1] + ax[2] * ax[2]);
    ax[0] /= nrm; ax[1] /= nrm; ax[2] /= nrm;                       /* :36 */
    REAL n2 = SQRT(ax[0] * ax[0] + ax[1] * ax[1] + ax[2] * ax[2]);  /* util.py:201 */
    REAL w[3] = {ax[0] / n2 * (REAL)ang, ax[1] / n2 * (REAL)ang, ax[2] / n2 * (REAL)ang};
    REAL r[9];
    FN(exp3)(w, r);
    if (mean) FN(mul33)(mean, r, out + 9 * b);                      /* :50 */
    else for (int i = 0; i < 9; i++) out[9 * b + i] = r[i];
  }
}

/* SinusoidalPosEmb (models.py:13-25) + RotPredict.forward (so3_train.py:39-49),
 * out_type="skewvec".  params = flat state_dict order:
 *   net.0.weight[65,65] net.0.bias[65] net.2.* net.4.* net.6.* net.8.weight[3,65] net.8.bias[3]
 * freqs[28] = exp(arange(28) * -(ln 1e4 / 27)) evaluated in fp32 (models.py:19-21).
 * Embedding angles t*freq are formed in fp32 (int64 -> fp32 promotion), as torch does.
 * acts (optional) receives the 4 post-SiLU layers + input, [n][5][65], for the backward. */
#define SO3O_D 65
static inline REAL FN(silu)(REAL x) { return x / (1 + EXP(-x)); }

static void FN(mlp_input)(const REAL* R, long t, const float* freqs, REAL* x) {
  for (int i = 0; i < 9; i++) x[i] = R[i];
  for (int i = 0; i < 28; i++) {
    float a = (float)t * freqs[i];
    x[9 + i] = (REAL)SIN((REAL)a);
    x[9 + 28 + i] = (REAL)COS((REAL)a);
  }
}

static inline void FN(mlp_fwd_one)(const float* params, const float* freqs, const REAL* R, long t,
                                   REAL* out, REAL* acts, int nout) {
  const int D = SO3O_D;
  REAL h[2][SO3O_D];
  FN(mlp_input)(R, t, freqs, h[0]);
  if (acts) for (int i = 0; i < D; i++) acts[i] = h[0][i];
  const float* p = params;
  int cur = 0;
  for (int l = 0; l < 4; l++) {
    const float* W = p; const float* bias = p + D * D; p += D * D + D;
    for (int o = 0; o < D; o++) {
      REAL acc = (REAL)bias[o];
      for (int i = 0; i < D; i++) acc += (REAL)W[o * D + i] * h[cur][i];
      h[1 - cur][o] = FN(silu)(acc);
    }
    cur = 1 - cur;
    if (acts) for (int i = 0; i < D; i++) acts[(l + 1) * D + i] = h[cur][i];
  }
  const float* W = p; const float* bias = p + nout * D;  /* nout = 3 ("skewvec") or 6 ("rotmat", so3_train.py:19-22) */
  for (int o = 0; o < nout; o++) {
    REAL acc = (REAL)bias[o];
    for (int i = 0; i < D; i++) acc += (REAL)W[o * D + i] * h[cur][i];
    out[o] = acc;
  }
}

void FN(mlp_fwd)(const float* params, const float* freqs, const REAL* R, const long* t, long t_stride,
                 REAL* out, REAL* acts, long n, long nout) {
#pragma omp parallel for schedule(static)
  for (long b = 0; b < n; b++)
    FN(mlp_fwd_one)(params, freqs, R + 9 * b, t[b * t_stride], out + nout * b, acts ? acts + b * 5 * SO3O_D : 0, (int)nout);
}

/* Backward of RotPredict for a given dL/dout [n,3]: gradients wrt the 17,358
 * parameters, flat state_dict order (autograd of so3_train.py:39-49; the
 * rotation inputs carry no grad for loss_type="skewvec").
 * Accumulates in double regardless of REAL so the oracle is the better-conditioned side. */
void FN(mlp_bwd)(const float* params, const float* freqs, const REAL* R, const long* t, long t_stride,
                 const REAL* dout, double* dparams, long n, long nout) {
  const int D = SO3O_D;
  const long NP = 4 * (D * D + D) + nout * D + nout;
  for (long i = 0; i < NP; i++) dparams[i] = 0.0;
  for (long b = 0; b < n; b++) {
    REAL h[5][SO3O_D], z[4][SO3O_D];
    FN(mlp_input)(R + 9 * b, t[b * t_stride], freqs, h[0]);
    const float* p = params;
    for (int l = 0; l < 4; l++) {
      const float* W = p; const float* bias = p + D * D; p += D * D + D;
      for (int o = 0; o < D; o++) {
        REAL acc = (REAL)bias[o];
        for (int i = 0; i < D; i++) acc += (REAL)W[o * D + i] * h[l][i];
        z[l][o] = acc; h[l + 1][o] = FN(silu)(acc);
      }
    }
    /* last layer */
    long off5 = 4 * (D * D + D);
    REAL dh[SO3O_D], dz[SO3O_D];
    for (int i = 0; i < D; i++) dh[i] = 0;
    for (int o = 0; o < nout; o++) {
      REAL g = dout[nout * b + o];
      dparams[off5 + nout * D + o] += (double)g;
      for (int i = 0; i < D; i++) {
        dparams[off5 + o * D + i] += (double)g * (double)h[4][i];
        dh[i] += (REAL)params[off5 + o * D + i] * g;
      }
    }
    for (int l = 3; l >= 0; l--) {
      long off = (long)l * (D * D + D);
      for (int o = 0; o < D; o++) {
        REAL sg = 1 / (1 + EXP(-z[l][o]));
        dz[o] = dh[o] * (sg * (1 + z[l][o] * (1 - sg)));
      }
      for (int i = 0; i < D; i++) dh[i] = 0;
      for (int o = 0; o < D; o++) {
        dparams[off + D * D + o] += (double)dz[o];
        for (int i = 0; i < D; i++) {
          dparams[off + o * D + i] += (double)dz[o] * (double)h[l][i];
          dh[i] += (REAL)params[off + o * D + i] * dz[o];
        }
      }
    }
  }
}

/* Forward noising + regression target: SO3Diffusion.q_sample (diffusion.py:339-346)
 * and the target of p_losses (diffusion.py:355):
 *   x_t = so3_scale(x_0, sqrt(abar_t)) @ noise ; target = vee(log noise) / eps_t */
void FN(q_sample_target)(const REAL* x0, const REAL* noise, const float* sqrt_ac, const float* sqrt_1mac,
                         const long* t, REAL* x_t, REAL* target, long n) {
  for (long b = 0; b < n; b++) {
    REAL w[3], xs[9];
    FN(log3)(x0 + 9 * b, w);
    REAL k = (REAL)sqrt_ac[t[b]];
    w[0] *= k; w[1] *= k; w[2] *= k;
    FN(exp3)(w, xs);
    FN(mul33)(xs, noise + 9 * b, x_t + 9 * b);
    if (target) {
      REAL lw[3];
      FN(log3)(noise + 9 * b, lw);
      REAL ie = (REAL)1 / (REAL)sqrt_1mac[t[b]];
      target[3 * b] = lw[0] * ie; target[3 * b + 1] = lw[1] * ie; target[3 * b + 2] = lw[2] * ie;
    }
  }
}

/* Reverse-step mean: predict_start_from_noise (diffusion.py:291-297) and
 * q_posterior (diffusion.py:299-306):
 *   x0hat = so3_scale(x, a) @ exp(hat(v*b))^T ;  mean = so3_scale(x0hat, c1) @ so3_scale(x, c2) */
static inline void FN(p_mean_one)(const REAL* x, const REAL* v, REAL a, REAL bcoef, REAL c1, REAL c2, REAL* x0hat, REAL* mean) {
  REAL w[3], wa[3], xa[9], nv[3], nt[9], xh[9], wh[3], e1[9], e2[9];
  FN(log3)(x, w);
  wa[0] = w[0] * a; wa[1] = w[1] * a; wa[2] = w[2] * a;
  FN(exp3)(wa, xa);
  nv[0] = v[0] * bcoef; nv[1] = v[1] * bcoef; nv[2] = v[2] * bcoef;
  FN(exp3)(nv, nt);
  FN(mul33_bt)(xa, nt, xh);
  if (x0hat) for (int i = 0; i < 9; i++) x0hat[i] = xh[i];
  FN(log3)(xh, wh);
  wh[0] *= c1; wh[1] *= c1; wh[2] *= c1;
  FN(exp3)(wh, e1);
  /* the reference recomputes log(x) (diffusion.py:301); same value */
  w[0] *= c2; w[1] *= c2; w[2] *= c2;
  FN(exp3)(w, e2);
  FN(mul33)(e1, e2, mean);
}

void FN(p_mean)(const REAL* x, const REAL* v, REAL a, REAL bcoef, REAL c1, REAL c2, REAL* x0hat, REAL* mean, long n) {
#pragma omp parallel for schedule(static)
  for (long b = 0; b < n; b++)
    FN(p_mean_one)(x + 9 * b, v + 3 * b, a, bcoef, c1, c2, x0hat ? x0hat + 9 * b : 0, mean + 9 * b);
}

/* right-multiply by a per-sample rotation: mean @ sample (diffusion.py:326) */
void FN(rmul)(const REAL* a, const REAL* b, REAL* out, long n) {
  for (long i = 0; i < n; i++) { REAL o[9]; FN(mul33)(a + 9 * i, b + 9 * i, o); for (int j = 0; j < 9; j++) out[9 * i + j] = o[j]; }
}

/* log_prob score: d log f(omega(R)) / dR, the autograd result of
 * distributions.py:74-77 / 189-190, via the analytic chain rule
 *   d omega / dR = [ c/(4s) (R - R^T) - (s/2) I ] / (s^2 + c^2)   
 * dlogf = f'(omega)/f(omega) supplied by the caller (so3o_igso3_dlogf). */
void FN(domega_dR)(const REAL* R, REAL* g, long n) {
  for (long b = 0; b < n; b++) {
    const REAL* r = R + 9 * b;
    REAL v0 = r[7] - r[5], v1 = -(r[6] - r[2]), v2 = r[3] - r[1];
    REAL s = SQRT(v0 * v0 + v1 * v1 + v2 * v2) / 2;
    REAL c = (r[0] + r[4] + r[8] - 1) / 2;
    REAL den = s * s + c * c;
    REAL k = c / (4 * s);
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++)
        g[9 * b + 3 * i + j] = (k * (r[3 * i + j] - r[3 * j + i]) - (i == j ? s / 2 : 0)) / den;
  }
}

/* ---- so3_lock_train.RotPredict (so3_lock_train.py:11-59): d_model = 255, six ResLayer(Linear+SiLU) blocks
 * (models.py:28-34: x + layer(x)), then Linear(255, 3) for out_type="skewvec".  Input = [R(9), sin(123), cos(123)]
 * (models.py:13-25 with dim = 246).  Params flat in state_dict order: 6 x (W[255][255], b[255]), W_out[3][255], b_out[3]. */
#define SO3O_DW 255
static void FN(resnet_input)(const REAL* R, long t, const float* freqs, REAL* x) {
  for (int i = 0; i < 9; i++) x[i] = R[i];
  for (int i = 0; i < 123; i++) {
    float a = (float)t * freqs[i];
    x[9 + i] = (REAL)SIN((REAL)a);
    x[9 + 123 + i] = (REAL)COS((REAL)a);
  }
}

void FN(resnet_fwd)(const float* params, const float* freqs, const REAL* R, const long* t, long t_stride, REAL* out, long n,
                    long nout) {
  const int D = SO3O_DW;
#pragma omp parallel for schedule(static)
  for (long b = 0; b < n; b++) {
    REAL x[SO3O_DW], y[SO3O_DW];
    FN(resnet_input)(R + 9 * b, t[b * t_stride], freqs, x);
    const float* p = params;
    for (int l = 0; l < 6; l++) {
      const float* W = p; const float* bias = p + D * D; p += D * D + D;
      for (int o = 0; o < D; o++) {
        REAL acc = (REAL)bias[o];
        for (int i = 0; i < D; i++) acc += (REAL)W[o * D + i] * x[i];
        y[o] = x[o] + FN(silu)(acc);
      }
      for (int o = 0; o < D; o++) x[o] = y[o];
    }
    for (int o = 0; o < nout; o++) {
      REAL acc = (REAL)p[nout * D + o];
      for (int i = 0; i < D; i++) acc += (REAL)p[o * D + i] * x[i];
      out[nout * b + o] = acc;
    }
  }
}

/* gradients wrt the 392,448 parameters for a given dL/dout [n,3] (autograd of so3_lock_train.py:50-59), accumulated in double */
void FN(resnet_bwd)(const float* params, const float* freqs, const REAL* R, const long* t, long t_stride, const REAL* dout,
                    double* dparams, long n, long nout) {
  const int D = SO3O_DW;
  const long LS = (long)D * D + D, NP = 6 * LS + nout * D + nout;
  for (long i = 0; i < NP; i++) dparams[i] = 0.0;
  for (long b = 0; b < n; b++) {
    REAL x[7][SO3O_DW], z[6][SO3O_DW], dx[SO3O_DW], dz[SO3O_DW];
    FN(resnet_input)(R + 9 * b, t[b * t_stride], freqs, x[0]);
    for (int l = 0; l < 6; l++) {
      const float* W = params + l * LS; const float* bias = W + D * D;
      for (int o = 0; o < D; o++) {
        REAL acc = (REAL)bias[o];
        for (int i = 0; i < D; i++) acc += (REAL)W[o * D + i] * x[l][i];
        z[l][o] = acc; x[l + 1][o] = x[l][o] + FN(silu)(acc);
      }
    }
    const long off6 = 6 * LS;
    for (int i = 0; i < D; i++) dx[i] = 0;
    for (int o = 0; o < nout; o++) {
      REAL g = dout[nout * b + o];
      dparams[off6 + nout * D + o] += (double)g;
      for (int i = 0; i < D; i++) {
        dparams[off6 + o * D + i] += (double)g * (double)x[6][i];
        dx[i] += (REAL)params[off6 + o * D + i] * g;
      }
    }
    for (int l = 5; l >= 0; l--) {
      const long off = l * LS;
      for (int o = 0; o < D; o++) {
        REAL sg = 1 / (1 + EXP(-z[l][o]));
        dz[o] = dx[o] * (sg * (1 + z[l][o] * (1 - sg)));
      }
      for (int o = 0; o < D; o++) {  /* dx_l = dx_{l+1} + W^T dz */
        dparams[off + D * D + o] += (double)dz[o];
        for (int i = 0; i < D; i++) {
          dparams[off + o * D + i] += (double)dz[o] * (double)x[l][i];
          dx[i] += (REAL)params[off + o * D + i] * dz[o];
        }
      }
    }
  }
}
#undef SO3O_DW

/* ---- rotation-matrix head and the "prevstep" loss ---------------------------------------------
 * six2rmat (util.py:67-76): Gram-Schmidt of two 3-vectors, rows b1, b2, b1 x b2. */
void FN(six2rmat)(const REAL* x, REAL* out, long n) {
  for (long b = 0; b < n; b++) {
    const REAL* a1 = x + 6 * b; const REAL* a2 = a1 + 3; REAL* o = out + 9 * b;
    REAL n1 = SQRT(a1[0] * a1[0] + a1[1] * a1[1] + a1[2] * a1[2]);
    REAL b1[3] = {a1[0] / n1, a1[1] / n1, a1[2] / n1};
    REAL d = b1[0] * a2[0] + b1[1] * a2[1] + b1[2] * a2[2];
    REAL u[3] = {a2[0] - d * b1[0], a2[1] - d * b1[1], a2[2] - d * b1[2]};
    REAL n2 = SQRT(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
    REAL b2[3] = {u[0] / n2, u[1] / n2, u[2] / n2};
    o[0] = b1[0]; o[1] = b1[1]; o[2] = b1[2]; o[3] = b2[0]; o[4] = b2[1]; o[5] = b2[2];
    o[6] = b1[1] * b2[2] - b1[2] * b2[1]; o[7] = b1[2] * b2[0] - b1[0] * b2[2]; o[8] = b1[0] * b2[1] - b1[1] * b2[0];
  }
}
/* autograd of six2rmat: G = dL/dout [n,3,3] -> dL/dx [n,6] */
void FN(six2rmat_bwd)(const REAL* x, const REAL* G, REAL* dx, long n) {
  for (long b = 0; b < n; b++) {
    const REAL* a1 = x + 6 * b; const REAL* a2 = a1 + 3; const REAL* g = G + 9 * b;
    REAL n1 = SQRT(a1[0] * a1[0] + a1[1] * a1[1] + a1[2] * a1[2]);
    REAL b1[3] = {a1[0] / n1, a1[1] / n1, a1[2] / n1};
    REAL d = b1[0] * a2[0] + b1[1] * a2[1] + b1[2] * a2[2];
    REAL u[3] = {a2[0] - d * b1[0], a2[1] - d * b1[1], a2[2] - d * b1[2]};
    REAL n2 = SQRT(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
    REAL b2[3] = {u[0] / n2, u[1] / n2, u[2] / n2};
    const REAL* g1 = g; const REAL* g2 = g + 3; const REAL* g3 = g + 6;
    /* b3 = b1 x b2:  db1 += b2 x g3,  db2 += g3 x b1 */
    REAL gb1[3] = {g1[0] + b2[1] * g3[2] - b2[2] * g3[1], g1[1] + b2[2] * g3[0] - b2[0] * g3[2], g1[2] + b2[0] * g3[1] - b2[1] * g3[0]};
    REAL gb2[3] = {g2[0] + g3[1] * b1[2] - g3[2] * b1[1], g2[1] + g3[2] * b1[0] - g3[0] * b1[2], g2[2] + g3[0] * b1[1] - g3[1] * b1[0]};
    REAL p2 = gb2[0] * b2[0] + gb2[1] * b2[1] + gb2[2] * b2[2];
    REAL du[3] = {(gb2[0] - p2 * b2[0]) / n2, (gb2[1] - p2 * b2[1]) / n2, (gb2[2] - p2 * b2[2]) / n2};
    REAL q = b1[0] * du[0] + b1[1] * du[1] + b1[2] * du[2];
    for (int k = 0; k < 3; k++) {
      dx[6 * b + 3 + k] = du[k] - q * b1[k];        /* u = a2 - (b1.a2) b1 */
      gb1[k] += -d * du[k] - q * a2[k];
    }
    REAL p1 = gb1[0] * b1[0] + gb1[1] * b1[1] + gb1[2] * b1[2];
    for (int k = 0; k < 3; k++) dx[6 * b + k] = (gb1[k] - p1 * b1[k]) / n1;
  }
}

/* d omega / dM and (s, c, omega) of a matrix, util.py:165-169 */
static inline REAL FN(domega)(const REAL* r, REAL* g, REAL* s_out) {
  REAL v0 = r[7] - r[5], v1 = -(r[6] - r[2]), v2 = r[3] - r[1];
  REAL s = SQRT(v0 * v0 + v1 * v1 + v2 * v2) / 2;
  REAL c = (r[0] + r[4] + r[8] - 1) / 2;
  REAL den = s * s + c * c, k = c / (4 * s);
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) g[3 * i + j] = (k * (r[3 * i + j] - r[3 * j + i]) - (i == j ? s / 2 : 0)) / den;
  *s_out = s;
  return ATAN2(s, c);
}

/* autograd of log_rmat (util.py:164-175, generic branch): log = scale (R - R^T), scale = omega / (2 s);
 * G = dL/dlog [n,3,3] -> dL/dR.  d scale = d omega / (2 s) - omega / (2 s^2) ds,  ds/dR = (R - R^T) / (4 s). */
void FN(log_rmat_bwd)(const REAL* R, const REAL* G, REAL* dR, long n) {
  for (long b = 0; b < n; b++) {
    const REAL* r = R + 9 * b; const REAL* g = G + 9 * b;
    REAL dom[9], s;
    REAL om = FN(domega)(r, dom, &s);
    REAL scale = om / (2 * s), gs = 0;
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) gs += g[3 * i + j] * (r[3 * i + j] - r[3 * j + i]);
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) {
        REAL S = r[3 * i + j] - r[3 * j + i];
        dR[9 * b + 3 * i + j] = scale * (g[3 * i + j] - g[3 * j + i]) + gs * (dom[3 * i + j] / (2 * s) - om / (2 * s * s) * S / (4 * s));
      }
  }
}

/* autograd of rmat_dist (util.py:315-322): dist = |log(a^T b)|_F = sqrt(2) omega(a^T b);  g = dL/ddist [n] */
void FN(rmat_dist_bwd)(const REAL* A, const REAL* Bm, const REAL* g, REAL* dA, REAL* dB, long n) {
  for (long b = 0; b < n; b++) {
    REAL M[9], dom[9], s;
    FN(mul33_at)(A + 9 * b, Bm + 9 * b, M);
    FN(domega)(M, dom, &s);
    const REAL k = SQRT((REAL)2) * g[b];
    for (int i = 0; i < 9; i++) dom[i] *= k;
    FN(mul33_bt)(Bm + 9 * b, dom, dA + 9 * b);  /* M = a^T b: dL/da = b dM^T */
    FN(mul33)(A + 9 * b, dom, dB + 9 * b);      /*            dL/db = a dM   */
  }
}

/* loss_type = "prevstep" (diffusion.py:358-365): posterior mean (299-302) relative to x_noisy, squared geodesic distance to
 * the network's rotation.  dist2[n] = rmat_dist(x_recon, step)^2 (the loss is its mean); dx = d(sum dist2)/dx_recon. */
void FN(prevstep_loss)(const REAL* x_recon, const REAL* x_start, const REAL* x_noisy, const float* coef1, const float* coef2,
                       const long* t, REAL* step_out, REAL* dist2, REAL* dx, long n) {
  for (long b = 0; b < n; b++) {
    REAL w[3], c1m[9], c2m[9], pm[9], step[9], M[9], dom[9], s;
    FN(log3)(x_start + 9 * b, w);
    for (int k = 0; k < 3; k++) w[k] *= (REAL)coef1[t[b]];
    FN(exp3)(w, c1m);
    FN(log3)(x_noisy + 9 * b, w);
    for (int k = 0; k < 3; k++) w[k] *= (REAL)coef2[t[b]];
    FN(exp3)(w, c2m);
    FN(mul33)(c1m, c2m, pm);
    FN(mul33_at)(x_noisy + 9 * b, pm, step);
    if (step_out) for (int k = 0; k < 9; k++) step_out[9 * b + k] = step[k];
    FN(mul33_at)(x_recon + 9 * b, step, M);
    REAL om = FN(domega)(M, dom, &s);
    dist2[b] = 2 * om * om;
    if (dx) {
      for (int k = 0; k < 9; k++) dom[k] *= 4 * om;
      FN(mul33_bt)(step, dom, dx + 9 * b);
    }
  }
}

#undef SO3O_D
#undef FN
#undef CAT
#undef CAT_
