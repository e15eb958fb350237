// so3x_math.hpp -- per-sample SO(3) device math shared by every kernel.
//
// One lane = one rotation held in 9 VGPRs (row-major).  The closed forms replace
// the reference's generic torch.matrix_exp / torch.svd (util.py:204,360,105):
//   log  : atan2 form of util.py:164-192 (+ skew2vec, util.py:79-84)
//   exp  : Rodrigues  I + A K + B K^2  for torch.matrix_exp(vec2skew(w))
// Everything is fp32 (the reference's working precision); no fast-math, NaN
// semantics of the reference (axis = 0/0 at angle 0) are kept where documented.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace so3x {

constexpr float kPi = 3.14159265358979323846f;

// sin and cos of an fp32 angle with a 3-term Cody-Waite reduction by pi/2 done with FMAs
// (the product k*hi is not rounded inside an fma), then the classic degree-7/8 minimax
// polynomials on [-pi/4, pi/4].  Absolute error ~1e-7 for |a| < 1e5 rad, which covers every
// angle on the path: time-embedding angles t*freq <= T, and Rodrigues angles up to
// pi*sqrt(1/abar_T) ~ 6.4e4 with the cosine schedule.  Replaces ocml's
// sincosf (whose Payne-Hanek slow path costs ~100 VGPRs once inlined 5x per sample-step).
__device__ __forceinline__ void sincos_cw(float a, float* sn, float* cs) {
  const float k = rintf(a * 0x1.45f306p-1f);
  float r = fmaf(-k, 0x1.921fb6p+0f, a);
  r = fmaf(-k, -0x1.777a5cp-25f, r);
  r = fmaf(-k, -0x1.ee59dap-50f, r);
  const int q = (int)k;
  const float z = r * r;
  const float S = fmaf(fmaf(fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f) * z, r, r);
  const float C = fmaf(fmaf(fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f) * z, z,
                       fmaf(-0.5f, z, 1.0f));
  const float s1 = (q & 1) ? C : S;
  const float c1 = (q & 1) ? S : C;
  *sn = (q & 2) ? -s1 : s1;
  *cs = ((q + 1) & 2) ? -c1 : c1;
}

// Hardware sine / cosine (v_sin_f32 / v_cos_f32: argument in REVOLUTIONS, 8 issue cycles each, against ~25 instructions
// = ~110 cycles for sincos_cw).  Measured absolute error <= 2e-6 on [0, 1) revolutions (tests/test_gpu_parity.py), so:
//  * used where 1e-6 is invisible -- the direction of a Philox noise axis (unit_axis, all kernels alike), and the five
//    exponentials of a reverse step in the bf16 chain kernels, whose step error is the bf16 network's 1e-3;
//  * NOT used on the fp32 parity paths (gates G1 / G2 are 1e-5 with a 1e-6 median).
// Large arguments (a * theta reaches 6.4e4 rad = 1e4 revolutions at t = T-1) are reduced with v_fract_f32: the product in
// revolutions carries half an ulp = the error the fp32 angle a * theta has already.
__device__ __forceinline__ void sincos_rev(float rev, float* sn, float* cs) {
  *sn = __builtin_amdgcn_sinf(rev);
  *cs = __builtin_amdgcn_cosf(rev);
}
template <bool FAST> __device__ __forceinline__ void sincos_sel(float a, float* sn, float* cs);
template <> __device__ __forceinline__ void sincos_sel<false>(float a, float* sn, float* cs) { sincos_cw(a, sn, cs); }
template <> __device__ __forceinline__ void sincos_sel<true>(float a, float* sn, float* cs) {
  sincos_rev(__builtin_amdgcn_fractf(a * 0.15915494309189535f), sn, cs);
}

// 1-ulp hardware reciprocal / sqrt / rsqrt.  The IEEE-exact division and sqrt sequences
// (v_div_scale/fmas/fixup, ~10 instructions each) cost more than the rest of the rotation
// math; 1 ulp (6e-8) is far inside the 1e-5 parity gates.  Semantics at 0 / inf are the
// IEEE ones (rcp(0) = inf), which the 0/0 -> NaN behaviours of the reference rely on.
__device__ __forceinline__ float frcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float fdiv(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }
__device__ __forceinline__ float fsqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
__device__ __forceinline__ float frsq(float x) { return __builtin_amdgcn_rsqf(x); }

// logistic function for the fp32 network paths from the hardware exp2 / rcp (1 ulp each; the -log2(e) x product adds
// |x| 2^-24 relative error to the exponential: ~1e-6 at |x| = 20, where the sigmoid has long saturated; parity gate G5 is
// 1e-5).  ocml's expf plus an IEEE division is ~30 instructions per element and made the fp32 network kernels VALU-bound.
__device__ __forceinline__ float sigmoid_f32(float x) {
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896341f * x));
}

// atan2(s, c) for s >= 0 (result in [0, pi]): odd minimax polynomial on [0, 1] with exact unit
// slope at 0 (max abs error 1.1e-7, relative accuracy kept for small angles), octant folding.
__device__ __forceinline__ float atan2_pos(float s, float c) {
  const float ac = fabsf(c);
  const float mx = fmaxf(s, ac), mn = fminf(s, ac);
  const float a = fdiv(mn, mx);  // 0/0 -> NaN only when s == c == 0 (not a rotation)
  const float z = a * a;
  float q = -0x1.1d7010p-8f;
  q = fmaf(q, z, 0x1.797dd0p-6f);
  q = fmaf(q, z, -0x1.d9485ep-5f);
  q = fmaf(q, z, 0x1.912c20p-4f);
  q = fmaf(q, z, -0x1.1e3d90p-3f);
  q = fmaf(q, z, 0x1.98d610p-3f);
  q = fmaf(q, z, -0x1.5550f2p-2f);
  float r = fmaf(a * z, q, a);
  r = s > ac ? 1.57079632679489662f - r : r;
  return c < 0.0f ? 3.14159265358979324f - r : r;
}

// the same for c >= 0 (result in [0, pi/2]): what a quaternion's angle needs -- atan2(|xyz|, |w|) -- without the dead c < 0 fold
__device__ __forceinline__ float atan2_pos_pos(float s, float c) {
  const float mx = fmaxf(s, c), mn = fminf(s, c);
  const float a = fdiv(mn, mx);
  const float z = a * a;
  float q = -0x1.1d7010p-8f;
  q = fmaf(q, z, 0x1.797dd0p-6f);
  q = fmaf(q, z, -0x1.d9485ep-5f);
  q = fmaf(q, z, 0x1.912c20p-4f);
  q = fmaf(q, z, -0x1.1e3d90p-3f);
  q = fmaf(q, z, 0x1.98d610p-3f);
  q = fmaf(q, z, -0x1.5550f2p-2f);
  const float r = fmaf(a * z, q, a);
  return s > c ? 1.57079632679489662f - r : r;
}

// vee(R - R^T)-based log, as a 3-vector.  util.py:164-192.
//   s = |v|/2, c = (tr R - 1)/2, angle = atan2(s, c), w = v * angle/(2 s);
//   angle == 0 -> 0 (util.py:174).  s == 0 with c < 0 (exact pi) is the reference's
//   eigh branch (util.py:178-191, which takes an eigenvector ROW -- a reference bug);
//   the correct axis from diag((R+I)/2) is used instead (parity unpinned there).
// (log3_sc also returns s = sin(angle) >= 0 and c = cos(angle) as read off the matrix)
__device__ __forceinline__ float log3_sc(const float* R, float* w, float* s_out, float* c_out) {
  float v0 = R[7] - R[5], v1 = R[2] - R[6], v2 = R[3] - R[1];
  float s = fsqrt(v0 * v0 + v1 * v1 + v2 * v2) * 0.5f;
  float c = (R[0] + R[4] + R[8] - 1.0f) * 0.5f;
  *s_out = s; *c_out = c;
  float ang = atan2_pos(s, c);
  float scale = ang * frcp(2.0f * s);
  if (ang == 0.0f) scale = 0.0f;
  w[0] = scale * v0; w[1] = scale * v1; w[2] = scale * v2;
  if (s == 0.0f && ang != 0.0f) {  // exact pi rotation: rare, divergent on purpose
    float d0 = (R[0] + 1.f) * 0.5f, d1 = (R[4] + 1.f) * 0.5f, d2 = (R[8] + 1.f) * 0.5f;
    float a0, a1, a2;
    if (d0 >= d1 && d0 >= d2) { a0 = sqrtf(d0); a1 = (R[1] + R[3]) / (4.f * a0); a2 = (R[2] + R[6]) / (4.f * a0); }
    else if (d1 >= d2)        { a1 = sqrtf(d1); a0 = (R[1] + R[3]) / (4.f * a1); a2 = (R[5] + R[7]) / (4.f * a1); }
    else                      { a2 = sqrtf(d2); a0 = (R[2] + R[6]) / (4.f * a2); a1 = (R[5] + R[7]) / (4.f * a2); }
    w[0] = ang * a0; w[1] = ang * a1; w[2] = ang * a2;
  }
  return ang;  // the rotation angle in [0, pi] (= |w| up to rounding)
}
__device__ __forceinline__ void log3(const float* R, float* w) {
  float s, c;
  log3_sc(R, w, &s, &c);
}

// exp(hat(w)); hat per util.py:87-92.  |w| reaches ~6e4 rad at t = T-1
// (sqrt_recip_alphas_cumprod = 20291), hence sincos_cw.
__device__ __forceinline__ void exp3(const float* w, float* R) {
  float x = w[0], y = w[1], z = w[2];
  float t2 = x * x + y * y + z * z;
  float th = fsqrt(t2);
  float sn, cs;
  sincos_cw(th, &sn, &cs);
  const float it = frcp(th);
  float A = sn * it;
  float B = (1.0f - cs) * it * it;
  if (th < 1e-3f) {  // series: also covers th == 0 (rcp(0) = inf above)
    A = 1.0f - t2 * (1.0f / 6.0f);
    B = 0.5f - t2 * (1.0f / 24.0f);
  }
  R[0] = 1.0f + B * (x * x - t2); R[1] = B * x * y - A * z;       R[2] = B * x * z + A * y;
  R[3] = B * x * y + A * z;       R[4] = 1.0f + B * (y * y - t2); R[5] = B * y * z - A * x;
  R[6] = B * x * z - A * y;       R[7] = B * y * z + A * x;       R[8] = 1.0f + B * (z * z - t2);
}

// exp(hat(axis * ang)) for a UNIT axis and ang in [0, pi] (noise rotations).
__device__ __forceinline__ void exp_axis_angle(const float* ax, float ang, float* R) {
  float sn, cs;
  sincos_cw(ang, &sn, &cs);
  float x = ax[0], y = ax[1], z = ax[2], C = 1.0f - cs;
  R[0] = 1.0f + C * (x * x - 1.0f); R[1] = C * x * y - sn * z;        R[2] = C * x * z + sn * y;
  R[3] = C * x * y + sn * z;        R[4] = 1.0f + C * (y * y - 1.0f); R[5] = C * y * z - sn * x;
  R[6] = C * x * z - sn * y;        R[7] = C * y * z + sn * x;        R[8] = 1.0f + C * (z * z - 1.0f);
}

__device__ __forceinline__ void mul33(const float* a, const float* b, float* o) {
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) o[3 * i + j] = a[3 * i] * b[j] + a[3 * i + 1] * b[3 + j] + a[3 * i + 2] * b[6 + j];
}
__device__ __forceinline__ void mul33_bt(const float* a, const float* b, float* o) {  // a @ b^T
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++)
      o[3 * i + j] = a[3 * i] * b[3 * j] + a[3 * i + 1] * b[3 * j + 1] + a[3 * i + 2] * b[3 * j + 2];
}
__device__ __forceinline__ void mul33_at(const float* a, const float* b, float* o) {  // a^T @ b
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) o[3 * i + j] = a[i] * b[j] + a[3 + i] * b[3 + j] + a[6 + i] * b[6 + j];
}

// quat_to_rmat, util.py:222-252
__device__ __forceinline__ void quat_to_rmat(const float* q, float* o) {
  float r = q[0], i = q[1], j = q[2], k = q[3];
  float ts = 2.0f / (r * r + i * i + j * j + k * k);
  o[0] = 1.f - ts * (j * j + k * k); o[1] = ts * (i * j - k * r);       o[2] = ts * (i * k + j * r);
  o[3] = ts * (i * j + k * r);       o[4] = 1.f - ts * (i * i + k * k); o[5] = ts * (j * k - i * r);
  o[6] = ts * (i * k - j * r);       o[7] = ts * (j * k + i * r);       o[8] = 1.f - ts * (i * i + j * j);
}

// ---------------------------------------------------------------- Philox4x32-10
// Counter-based RNG: (seed) key, (sample index, stream offset) counter -> 4 x u32.
// Results depend only on (seed, global sample index, offset): identical for any
// launch geometry or number of GPUs.
struct Philox4 { uint32_t x, y, z, w; };

__device__ __forceinline__ Philox4 philox4x32_10(uint64_t seed, uint64_t ctr_lo, uint64_t ctr_hi) {
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
  uint32_t c0 = (uint32_t)ctr_lo, c1 = (uint32_t)(ctr_lo >> 32), c2 = (uint32_t)ctr_hi, c3 = (uint32_t)(ctr_hi >> 32);
#pragma unroll
  for (int r = 0; r < 10; r++) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    // a ^ b ^ c as ONE v_bitop3_b32 (truth table 0x96; gfx950): hipcc emits two v_xor for the plain expression
    uint32_t n0 = __builtin_amdgcn_bitop3_b32((uint32_t)(p1 >> 32), c1, k0, 0x96);
    uint32_t n2 = __builtin_amdgcn_bitop3_b32((uint32_t)(p0 >> 32), c3, k1, 0x96);
    uint32_t n1 = (uint32_t)p1;
    uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  return Philox4{c0, c1, c2, c3};
}

// [0,1) with 24 random bits, the granularity of torch.rand for fp32 (distributions.py:38)
__device__ __forceinline__ float u01(uint32_t u) { return (float)(u >> 8) * (1.0f / 16777216.0f); }

// Uniform direction on S^2 from two uniforms: distributionally identical to the
// reference's normalised Gaussian 3-vector (distributions.py:35-36) at 2 draws, not 3.
__device__ __forceinline__ void unit_axis(uint32_t a, uint32_t b, float* ax) {
  float z = 2.0f * u01(a) - 1.0f + (1.0f / 16777216.0f);  // centred: z in (-1, 1)
  float r = fsqrt(fmaxf(0.0f, 1.0f - z * z));
  float sn, cs;
  sincos_rev(u01(b), &sn, &cs);  // azimuth 2 pi u: the uniform IS the angle in revolutions
  ax[0] = r * cs; ax[1] = r * sn; ax[2] = z;
}

// Inverse-CDF angle, distributions.py:39-49, on one CDF row (999 fp32, global or LDS)
// and the 1000 knot angles.  wrow = the row the interpolation weight is gathered from
// (== row unless the column-0 quirk is on).
// EXACT = true keeps the reference's IEEE division in the interpolation weight (bit-identical angles on the
// reference's own CDF rows, used by every explicit-draw / parity path); false = 1-ulp reciprocal (in-kernel
// Philox paths, where no bitwise comparison with the reference is possible anyway).
// guide (optional, so3x_igso3_build_guide): guide[b] = #{k : row[k] <= b / 256}, b = 0..256, brackets the answer for every
// u of bin b, so the bisection starts ~4 knots wide instead of 999: with per-sample rows (training: t differs per lane)
// every probe is its own 128-byte L2 line, and 10 of them per sample made the search L2-bandwidth-bound.
constexpr int kGuideBins = 256, kGuidePitch = 258;  // = SO3X_GUIDE_BINS / SO3X_GUIDE_PITCH (checked in so3x_igso3.hip)
template <bool EXACT = true>
__device__ __forceinline__ float igso3_angle(const float* row, const float* wrow, const float* knots, float u,
                                             const uint16_t* guide = nullptr) {
  int lo = 0, hi = 999;  // idx1 = #{k : row[k] <= u}  (row is non-decreasing)
  if (guide) {
    int b = (int)(u * (float)kGuideBins);  // exact: power-of-two scale
    b = b < 0 ? 0 : (b > kGuideBins - 1 ? kGuideBins - 1 : b);
    const uint32_t g2 = *reinterpret_cast<const uint32_t*>(guide + (b & ~1));  // guide[b & ~1], guide[(b & ~1) + 1] in one load
    const uint32_t g3 = guide[b + 1];
    lo = (b & 1) ? (int)(g2 >> 16) : (int)(g2 & 0xffffu);
    hi = (b & 1) ? (int)g3 : (int)(g2 >> 16);
  }
#pragma unroll 1
  while (lo < hi) {
    int mid = (lo + hi) >> 1;
    if (row[mid] <= u) lo = mid + 1; else hi = mid;
  }
  int idx1 = lo > 998 ? 998 : lo;  // row[998] == 1 > u, so this clamp never bites
  int idx0 = idx1 - 1 < 0 ? 0 : idx1 - 1;
  float ts = wrow[idx0], te = wrow[idx1];
  float df = fmaxf(te - ts, 1e-6f);
  float wt = fminf(fmaxf(EXACT ? (u - ts) / df : (u - ts) * frcp(df), 0.0f), 1.0f);
  float a0 = knots[idx0 + 1], a1 = knots[idx1 + 1];
  float dl = a1 - a0;
  return wt < 0.5f ? a0 + wt * dl : a1 - dl * (1.0f - wt);  // torch.lerp's two-sided form
}

// The same search for rows in GLOBAL memory with a guide (the noising kernels: every lane has its own timestep, so every probe of
// the bisection above is a dependent L2 round trip on a line of its own, and a wave pays the rounds of its WORST lane: the
// guide leaves a bracket of <= 4 knots for 93 % of (row, u) pairs, but 3 % are wider than 8 and 1.5 % wider than 15 -- the flat
// tails of a CDF put hundreds of knots into one 1/256 bin -- so most 64-lane waves went 6-9 rounds).  Here the dependent
// chain is [guide: one 8-byte load] -> [window: the 8 knots around the bracket, two 16-byte loads issued together] for 96 %
// of the lanes; a wide bracket first shrinks 9x per round with 8 INDEPENDENT probes (856 -> 96 -> 11 -> 2: three rounds at most).
// idx1 = #{k : row[k] <= u} exactly as the bisection counts it (the row is non-decreasing and the guide's bracket is exact), so the
// angle is bit-identical; the interpolation ends come out of the window's registers when the weight row is the search row.
template <bool EXACT = true>
__device__ __forceinline__ float igso3_angle_windowed(const float* __restrict__ row, const float* __restrict__ wrow,
                                                      const float* __restrict__ knots, float u, const uint16_t* __restrict__ guide) {
  int b = (int)(u * (float)kGuideBins);
  b = b < 0 ? 0 : (b > kGuideBins - 1 ? kGuideBins - 1 : b);
  uint64_t g4;  // guide[b & ~1 .. (b & ~1) + 3]: four bytes aligned, one load
  __builtin_memcpy(&g4, guide + (b & ~1), 8);
  const int sh = (b & 1) * 16;
  int lo = (int)((g4 >> sh) & 0xffffu), hi = (int)((g4 >> (sh + 16)) & 0xffffu);
  hi = hi > 998 ? 998 : hi;  // row[998] == 1 > u: the count never exceeds 998
  while (hi - lo > 6) {      // rare per lane (3 %); 8 independent probes cut the bracket to a ninth
    const int step = (hi - lo + 8) / 9;
    float pv[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const int p = lo + (j + 1) * step;
      pv[j] = row[p > 998 ? 998 : p];
    }
    int c = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) c += (pv[j] <= u && lo + (j + 1) * step < hi) ? 1 : 0;  // monotone: a prefix
    const int nlo = c > 0 ? lo + c * step + 1 : lo;
    const int nhi = (c < 8 && lo + (c + 1) * step < hi) ? lo + (c + 1) * step : hi;
    lo = nlo; hi = nhi;
  }
  // window of 8 knots from wb = lo - 1 (so that row[idx1 - 1] is in it), clamped to the row: positions < lo are <= u and
  // positions >= hi are > u by the bracket's invariant, hi <= wb + 7, so idx1 = wb + #{window positions <= u} and
  // row[idx1], row[idx1 - 1] are window entries
  int wb = lo - 1 < 0 ? 0 : lo - 1;
  wb = wb > 991 ? 991 : wb;
  float win[8];
  __builtin_memcpy(win, row + wb, 32);
  int cnt = 0;
#pragma unroll
  for (int j = 0; j < 8; j++) cnt += (wb + j < lo || (wb + j < hi && win[j] <= u)) ? 1 : 0;
  const int idx1 = wb + cnt;                      // <= hi <= 998
  const int idx0 = idx1 - 1 < 0 ? 0 : idx1 - 1;
  float ts, te;
  if (wrow == row) {
    te = win[7]; ts = win[6];
#pragma unroll
    for (int j = 6; j >= 0; j--) {
      te = cnt == j ? win[j] : te;
      ts = cnt == j + 1 ? win[j] : ts;
    }
    ts = cnt == 0 ? win[0] : ts;                  // idx1 == 0: idx0 == idx1
  } else {
    ts = wrow[idx0]; te = wrow[idx1];             // the column-0 quirk: one shared row, hot in L1
  }
  const float df = fmaxf(te - ts, 1e-6f);
  const float wt = fminf(fmaxf(EXACT ? (u - ts) / df : (u - ts) * frcp(df), 0.0f), 1.0f);
  const float a0 = knots[idx0 + 1], a1 = knots[idx1 + 1];
  const float dl = a1 - a0;
  return wt < 0.5f ? a0 + wt * dl : a1 - dl * (1.0f - wt);
}

// rows in global memory: the windowed search when a guide exists, the plain bisection otherwise (same result)
template <bool EXACT = true>
__device__ __forceinline__ float igso3_angle_global(const float* __restrict__ row, const float* __restrict__ wrow,
                                                    const float* __restrict__ knots, float u, const uint16_t* __restrict__ guide) {
  return guide ? igso3_angle_windowed<EXACT>(row, wrow, knots, u, guide) : igso3_angle<EXACT>(row, wrow, knots, u, nullptr);
}

// ---------------------------------------------------------------- unit quaternions
// The chain-resident sampler keeps its state as a unit quaternion (w, x, y, z) between the steps of one launch:
// composition is 16 multiplies instead of 27, exp is a half-angle sincos, log is one atan2 with no 1/(pi - w)
// conditioning problem, and the matrix is only formed as the network's input (quat_to_rmat, util.py:222-252).
// hat() of the reference (util.py:87-92) is the standard cross-product matrix, so exp(hat(n * th)) <-> (cos th/2, n sin th/2).
struct Quat { float w, x, y, z; };

__device__ __forceinline__ Quat qmul(const Quat& a, const Quat& b) {  // R(a) R(b) = R(a (x) b)
  return Quat{a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z, a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
              a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x, a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w};
}
__device__ __forceinline__ Quat qnormalize(const Quat& q) {
  const float r = frsq(q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z);
  return Quat{q.w * r, q.x * r, q.y * r, q.z * r};
}
__device__ __forceinline__ void rmat_from_quat(const Quat& q, float* o) {  // unit q
  const float xx = q.x * q.x, yy = q.y * q.y, zz = q.z * q.z, xy = q.x * q.y, xz = q.x * q.z, yz = q.y * q.z;
  const float wx = q.w * q.x, wy = q.w * q.y, wz = q.w * q.z;
  o[0] = 1.f - 2.f * (yy + zz); o[1] = 2.f * (xy - wz);       o[2] = 2.f * (xz + wy);
  o[3] = 2.f * (xy + wz);       o[4] = 1.f - 2.f * (xx + zz); o[5] = 2.f * (yz - wx);
  o[6] = 2.f * (xz - wy);       o[7] = 2.f * (yz + wx);       o[8] = 1.f - 2.f * (xx + yy);
}
__device__ __forceinline__ Quat quat_from_rmat(const float* R) {  // Shepperd's branch choice, then normalised
  const float tr = R[0] + R[4] + R[8];
  Quat q;
  if (tr > 0.f) {
    const float s = 2.f * sqrtf(tr + 1.f);
    q = Quat{0.25f * s, (R[7] - R[5]) / s, (R[2] - R[6]) / s, (R[3] - R[1]) / s};
  } else if (R[0] > R[4] && R[0] > R[8]) {
    const float s = 2.f * sqrtf(1.f + R[0] - R[4] - R[8]);
    q = Quat{(R[7] - R[5]) / s, 0.25f * s, (R[1] + R[3]) / s, (R[2] + R[6]) / s};
  } else if (R[4] > R[8]) {
    const float s = 2.f * sqrtf(1.f + R[4] - R[0] - R[8]);
    q = Quat{(R[2] - R[6]) / s, (R[1] + R[3]) / s, 0.25f * s, (R[5] + R[7]) / s};
  } else {
    const float s = 2.f * sqrtf(1.f + R[8] - R[0] - R[4]);
    q = Quat{(R[3] - R[1]) / s, (R[2] + R[6]) / s, (R[5] + R[7]) / s, 0.25f * s};
  }
  return qnormalize(q);
}
// rotation angle in [0, pi] and unit axis (zero vector at the identity, where the angle is 0 anyway)
__device__ __forceinline__ float quat_axis_angle(const Quat& q, float* ax) {
  // q and -q are the same rotation: take w >= 0, i.e. the axis carries the sign of w (one bit-field insert) and the angle |w|
  const float n = fsqrt(q.x * q.x + q.y * q.y + q.z * q.z);
  const float inv = n > 0.f ? __builtin_copysignf(frcp(n), q.w) : 0.f;
  ax[0] = q.x * inv; ax[1] = q.y * inv; ax[2] = q.z * inv;
  return 2.f * atan2_pos_pos(n, fabsf(q.w));
}
template <bool FAST = false>
__device__ __forceinline__ Quat quat_axis_angle_exp(const float* ax, float ang) {
  float sn, cs;
  if constexpr (FAST) sincos_rev(__builtin_amdgcn_fractf(ang * 0.07957747154594767f), &sn, &cs);  // half the angle, in revolutions: one multiply
  else sincos_sel<false>(0.5f * ang, &sn, &cs);
  return Quat{cs, sn * ax[0], sn * ax[1], sn * ax[2]};
}

// ---------------------------------------------------------------- reverse-step mean
// reverse mean of one sample: predict_start_from_noise (diffusion.py:291-297) then
// q_posterior (299-302).  log(x) is evaluated once (the reference does it twice, 292 & 301).
__device__ __forceinline__ void p_mean_one(const float* x, const float* v, float a, float b, float c1, float c2,
                                           float* x0hat, float* mean) {
  float w[3], wa[3], xa[9], nv[3], nt[9], wh[3], e1[9], e2[9];
  log3(x, w);
  wa[0] = w[0] * a; wa[1] = w[1] * a; wa[2] = w[2] * a;
  exp3(wa, xa);
  nv[0] = v[0] * b; nv[1] = v[1] * b; nv[2] = v[2] * b;
  exp3(nv, nt);
  mul33_bt(xa, nt, x0hat);
  log3(x0hat, wh);
  wh[0] *= c1; wh[1] *= c1; wh[2] *= c1;
  exp3(wh, e1);
  w[0] *= c2; w[1] *= c2; w[2] *= c2;
  exp3(w, e2);
  mul33(e1, e2, mean);
}

// standard normals from two uniforms (Box-Muller); u in [0,1) from u01: shift to (0,1] for the log.  The azimuth 2 pi u2 IS the
// angle in revolutions, so it goes to the hardware sine / cosine as it is (2 instructions against ~25 for the Cody-Waite form; 2e-6
// absolute on a unit circle -- these are in-kernel Philox draws, pinned distributionally, never bitwise; round 4: 40.4 -> 39.4 us
// for k_se3_q_sample_target at 2^20 frames, profiles/r04_ab_se3_qsample.json)
__device__ __forceinline__ void box_muller(uint32_t a, uint32_t b, float* z0, float* z1) {
  const float u1 = u01(a) + (1.0f / 16777216.0f), u2 = u01(b);
  const float r = fsqrt(-2.0f * __logf(u1));
  float sn, cs;
  sincos_rev(u2, &sn, &cs);
  *z0 = r * cs; *z1 = r * sn;
}

}  // namespace so3x
