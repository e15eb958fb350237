// so3x_igso3.hip -- isotropic Gaussian on SO(3): density, CDF tables, inverse-CDF
// sampling, log-prob + score (reference distributions.py:8-81).
#include "so3x_common.hpp"
#include "so3x_math.hpp"
#include "so3x_igso3.hpp"

using namespace so3x;

namespace {

// ------------------------------------------------------------------ A1 pointwise
__global__ void __launch_bounds__(kBlock)
k_eps_ft(const float* __restrict__ omega, const float* __restrict__ eps, int64_t eps_stride, float* __restrict__ out, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock)
    out[i] = (float)eps_ft_f64((double)omega[i], (double)eps[i * eps_stride]);
}

// ------------------------------------------------------------------ A2 table build
// One 256-thread block per CDF row.  fp64 pdf at the 1000 knots, fp32 Haar product
// and trapezoid terms, double-accumulated prefix sum (torch's CPU cumsum accumulates
// fp32 in double; pinned by the goldens), fp32 normalisation.  distributions.py:15-30.
__global__ void __launch_bounds__(kBlock) k_build_tables(const float* __restrict__ eps, float* __restrict__ trap) {
  __shared__ float pdf[1024];
  __shared__ double wave_tot[4];
  __shared__ float s_last;
  const int row = blockIdx.x;
  const double e = (double)eps[row];
  for (int k = threadIdx.x; k < 1000; k += kBlock) {
    float f = (float)eps_ft_f64((double)SO3X_KNOTS_DATA[k], e);   // .float() at :72
    float p = f * SO3X_HAAR_W_DATA[k];                             // fp32 product, :21
    if (SO3X_KNOTS_DATA[k] == 0.0f) p = 0.0f;                      // :23
    pdf[k] = p;
  }
  __syncthreads();
  // each thread owns 4 consecutive trapezoid terms (999 = 4*249 + 3)
  double loc[4];
  double run = 0.0;
#pragma unroll
  for (int j = 0; j < 4; j++) {
    int k = 4 * threadIdx.x + j;
    float term = 0.0f;
    if (k < 999) {
      float sum = pdf[k] + pdf[k + 1];                             // :26
      float dl = SO3X_KNOTS_DATA[k + 1] - SO3X_KNOTS_DATA[k];      // :27
      term = dl * sum / 2.0f;                                      // :28
    }
    run += (double)term;
    loc[j] = run;
  }
  // inclusive scan of the per-thread totals: wave shuffles, then 4 wave totals via LDS
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  double incl = run;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    double o = __shfl_up(incl, d);
    if (lane >= d) incl += o;
  }
  if (lane == 63) wave_tot[wid] = incl;
  __syncthreads();
  double pre = incl - run;
  for (int w = 0; w < wid; w++) pre += wave_tot[w];
  float vals[4];
#pragma unroll
  for (int j = 0; j < 4; j++) vals[j] = (float)(pre + loc[j]);     // cumsum output cast, :28
  if (threadIdx.x == 249) s_last = vals[2];                         // k = 998
  __syncthreads();
  const float last = s_last;
#pragma unroll
  for (int j = 0; j < 4; j++) {
    int k = 4 * threadIdx.x + j;
    if (k < 999) trap[(int64_t)row * 999 + k] = vals[j] / last;    // :29
  }
}

// ------------------------------------------------------------------ A2 sampling
__global__ void __launch_bounds__(kBlock)
k_igso3_sample(const float* __restrict__ trap, const uint16_t* __restrict__ guide, const int64_t* __restrict__ row_idx,
               int64_t row_const, int quirk_col0,
               const float* __restrict__ axes, const float* __restrict__ unif, uint64_t seed, uint64_t rng_offset,
               int64_t index_base, const float* __restrict__ mean, float* __restrict__ out, float* __restrict__ angle_out,
               float* __restrict__ axis_out, int64_t n) {
  __shared__ __attribute__((aligned(16))) float sm[kTile * 9];
  const int64_t ntiles = (n + kTile - 1) / kTile;
  const int64_t wrow_i = (quirk_col0 && row_idx) ? row_idx[0] : -1;
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t base = tile * kTile;
    const int cnt = (int)((n - base) < kTile ? (n - base) : kTile);
    const int64_t idx = base + threadIdx.x;
    const bool live = threadIdx.x < cnt;
    float ax[3], u;
    if (axes) {
      float a[3];
      load_rows<3>(axes, base, cnt, sm, a);
      float nrm = sqrtf(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]);   // distributions.py:36
      ax[0] = a[0] / nrm; ax[1] = a[1] / nrm; ax[2] = a[2] / nrm;
      float n2 = sqrtf(ax[0] * ax[0] + ax[1] * ax[1] + ax[2] * ax[2]);  // util.py:201 renormalises
      ax[0] /= n2; ax[1] /= n2; ax[2] /= n2;
      u = live ? unif[idx] : 0.5f;
    } else {
      Philox4 r = philox4x32_10(seed, (uint64_t)(index_base + idx), rng_offset);
      unit_axis(r.x, r.y, ax);
      u = u01(r.z);
    }
    const int64_t ri = live ? (row_idx ? row_idx[idx] : row_const) : (row_idx ? row_idx[base] : row_const);
    const float* row = trap + ri * 999;
    const float* wrow = wrow_i >= 0 ? trap + wrow_i * 999 : row;
    float ang = igso3_angle_global(row, wrow, SO3X_KNOTS_DATA, u, guide ? guide + ri * kGuidePitch : nullptr);
    float r9[9], o[9];
    exp_axis_angle(ax, ang, r9);
    if (mean) {
      float m[9];
#pragma unroll
      for (int j = 0; j < 9; j++) m[j] = mean[j];
      mul33(m, r9, o);                                              // distributions.py:50
    } else {
#pragma unroll
      for (int j = 0; j < 9; j++) o[j] = r9[j];
    }
    if (angle_out && live) angle_out[idx] = ang;
    if (axis_out) store_rows<3>(axis_out, base, cnt, sm, ax);
    store_rows<9>(out, base, cnt, sm, o);
  }
}

// search guide of a CDF row: guide[b] = #{k : row[k] <= b / 256}, by the same bisection the samplers use
static_assert(kGuideBins == SO3X_GUIDE_BINS && kGuidePitch == SO3X_GUIDE_PITCH, "guide geometry");
__global__ void __launch_bounds__(kGuidePitch + 62) k_build_guide(const float* __restrict__ trap, uint16_t* __restrict__ guide) {
  const float* row = trap + (int64_t)blockIdx.x * 999;
  const int b = threadIdx.x;
  if (b >= kGuidePitch) return;
  const float u = (float)b * (1.0f / (float)kGuideBins);
  int lo = 0, hi = 999;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (row[mid] <= u) lo = mid + 1; else hi = mid;
  }
  guide[(int64_t)blockIdx.x * kGuidePitch + b] = (uint16_t)(b <= kGuideBins ? lo : 999);
}

// exact-identity input: the reference's fp64 limit expression; out of line so the fp64 code does not
// inflate the streaming kernel's register budget
__device__ __attribute__((noinline)) float logp_identity_f64(float e) {
  double dlogf;
  return logf((float)eps_ft_and_dlog_f64(0.0, (double)e, &dlogf));
}

// ------------------------------------------------------------------ A3 log-prob + score
// 56 algorithmic bytes per evaluation (36 R + 4 eps in, 4 logp + 12 score out): HBM-bound
// by design.  The reference evaluates the density in fp64, casts to fp32 and logs in fp32
// (distributions.py:53-77); igso3_logf_dlog_f32 reproduces that result in fp32 arithmetic
// (so3x_igso3.hpp) -- the fp64 form was VALU-bound at 27 % of HBM peak.
#ifndef SO3X_LPS_NT
#define SO3X_LPS_NT 1
#endif
// Waves per SIMD asked of the instantiation without the dense gradient: 6 (80 registers, no spill).  Measured on one box
// (tools/ab/ab_logprob.py), 2^20 / 2^24 evaluations: generic kernel at 5 waves 61 % / 72 % of 8 TB/s; this instantiation at
// 5: 64.2 / 72.6, at 6: 66.4 / 77.6, at 7: 65.7 / 77.5, at 8 (16 B of scratch): 62.0 / 71.6.
#ifndef SO3X_LPS_OCC
#define SO3X_LPS_OCC 6
#endif
// Non-temporal STORES of the outputs: +1-3 % (2^20: 60.4 -> 61.2 % of 8 TB/s, 2^24: 69.9 -> 71.9 %, tools/ab/ab_logprob.py).
// Non-temporal LOADS of the inputs: -8 % -- back-to-back calls on the same rotations find them in the memory-side cache.
#ifndef SO3X_LPS_NTL
#define SO3X_LPS_NTL 0
#endif
#if SO3X_LPS_NTL
#define SO3X_LPS_LOAD(p) __builtin_nontemporal_load(p)
#else
#define SO3X_LPS_LOAD(p) (*(p))
#endif
#if SO3X_LPS_NT
#define SO3X_LPS_STORE(v, p) __builtin_nontemporal_store(v, p)
#else
#define SO3X_LPS_STORE(v, p) (*(p) = (v))
#endif
template <bool GRAD>  // GRAD: the dense d logp / dR output as well (rare); without it the kernel is its own, leaner instantiation
__global__ void __launch_bounds__(kBlock, GRAD ? 5 : SO3X_LPS_OCC)
k_logprob_score(const float* __restrict__ R, const float* __restrict__ eps, int64_t eps_stride, float* __restrict__ logp,
                float* __restrict__ score_vec, float* __restrict__ grad_R_, int64_t n) {
  float* grad_R = GRAD ? grad_R_ : nullptr;
  // wave-private staging: every wave streams its own 64-sample tiles, no workgroup barrier
  __shared__ __attribute__((aligned(16))) float sm[kBlock / kWave][kWave * 9];
  float* wl = sm[threadIdx.x >> 6];
  const int lane = threadIdx.x & 63;
  const int64_t ntiles = (n + kWave - 1) / kWave;
  const int64_t wave = (int64_t)blockIdx.x * (kBlock / kWave) + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * (kBlock / kWave);
  // software prefetch: the next tile's loads are in flight while the current tile is computed
  const bool aligned = (reinterpret_cast<uintptr_t>(R) & 15) == 0;
  auto full = [&](int64_t tl) { return aligned && tl < ntiles && (n - tl * kWave) >= kWave; };
  Pref9 pf;
  float pe = 1.0f;
  bool have = full(wave);
  if (have) { pf = wave_prefetch9<SO3X_LPS_NTL>(R, wave * kWave); pe = SO3X_LPS_LOAD(eps + (wave * kWave + lane) * eps_stride); }
  for (int64_t tile = wave; tile < ntiles; tile += nwaves) {
    const int64_t base = tile * kWave;
    const int cnt = (int)((n - base) < kWave ? (n - base) : kWave);
    const int64_t idx = base + lane;
    const bool live = lane < cnt;
    float r[9], w[3];
    float e;
    if (have) {
      wave_commit9(pf, wl, r);
      e = pe;
    } else {
      e = live ? eps[idx * eps_stride] : 1.0f;
      wave_load_rows<9>(R, base, cnt, wl, r);
    }
    have = full(tile + nwaves);
    if (have) { pf = wave_prefetch9<SO3X_LPS_NTL>(R, (tile + nwaves) * kWave); pe = SO3X_LPS_LOAD(eps + ((tile + nwaves) * kWave + lane) * eps_stride); }
    float sn_om, cs_om;
    const float ang = log3_sc(r, w, &sn_om, &cs_om);   // rmat_to_aa angle (util.py:217): |w| = atan2(s, c)
    float lp, dl;
    if (ang == 0.0f) {  // exact identity: the reference's fp64 limit expression (rare, divergent on purpose)
      lp = logp_identity_f64(e);
      dl = 0.0f;
    } else {
      lp = igso3_logf_dlog_f32(ang, e, sn_om, cs_om, &dl);
    }
    if (live) SO3X_LPS_STORE(lp, logp + idx);
    if (score_vec) {
      const float k = dl * frcp(ang);
      float sv[3] = {k * w[0], k * w[1], k * w[2]};
      wave_store_rows<3, SO3X_LPS_NT>(score_vec, base, cnt, wl, sv);
    }
    if (grad_R) {
      // d omega / dR = [ c/(4s) (R - R^T) - (s/2) I ] / (s^2 + c^2)   
      float v0 = r[7] - r[5], v1 = r[2] - r[6], v2 = r[3] - r[1];
      float s = fsqrt(v0 * v0 + v1 * v1 + v2 * v2) * 0.5f;
      float c = (r[0] + r[4] + r[8] - 1.0f) * 0.5f;
      float inv = dl * frcp(s * s + c * c);
      float k = c * frcp(4.0f * s);
      float g[9];
#pragma unroll
      for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) g[3 * i + j] = (k * (r[3 * i + j] - r[3 * j + i]) - (i == j ? 0.5f * s : 0.0f)) * inv;
      wave_store_rows<9>(grad_R, base, cnt, wl, g);
    }
  }
}

}  // namespace

extern "C" {

int so3x_igso3_eps_ft(so3x_stream_t s, const float* omega, const float* eps, int64_t eps_stride, float* out, int64_t n) {
  if (n < 0 || (n && (!omega || !eps || !out)) || (eps_stride != 0 && eps_stride != 1)) return SO3X_ERR_INVALID_ARG;
  if (n == 0) return SO3X_OK;
  hipLaunchKernelGGL(k_eps_ft, dim3(grid_for_tiles((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, (hipStream_t)s, omega, eps,
                     eps_stride, out, n);
  return check_launch();
}

int so3x_igso3_build_tables(so3x_stream_t s, const float* eps, int64_t n_rows, float* trap) {
  if (n_rows < 0 || n_rows > 0x7fffffff || (n_rows && (!eps || !trap))) return SO3X_ERR_INVALID_ARG;
  if (n_rows == 0) return SO3X_OK;
  hipLaunchKernelGGL(k_build_tables, dim3((unsigned)n_rows), dim3(kBlock), 0, (hipStream_t)s, eps, trap);
  return check_launch();
}

int so3x_igso3_build_guide(so3x_stream_t s, const float* trap, int64_t n_rows, uint16_t* guide) {
  if (n_rows < 0 || n_rows > 0x7fffffff || (n_rows && (!trap || !guide))) return SO3X_ERR_INVALID_ARG;
  if (n_rows == 0) return SO3X_OK;
  hipLaunchKernelGGL(k_build_guide, dim3((unsigned)n_rows), dim3(kGuidePitch + 62), 0, (hipStream_t)s, trap, guide);
  return check_launch();
}

int so3x_igso3_sample(so3x_stream_t s, const float* trap, const uint16_t* guide, const int64_t* row_idx, int64_t row_const,
                      int quirk_col0,
                      const float* axes, const float* unif, uint64_t seed, uint64_t rng_offset, int64_t index_base,
                      const float* mean, float* out, float* angle_out, float* axis_out, int64_t n) {
  if (n < 0 || (n && (!trap || !out)) || ((axes == nullptr) != (unif == nullptr)) || row_const < 0)
    return SO3X_ERR_INVALID_ARG;
  if (n == 0) return SO3X_OK;
  hipLaunchKernelGGL(k_igso3_sample, dim3(grid_for_tiles((n + kTile - 1) / kTile)), dim3(kBlock), 0, (hipStream_t)s, trap,
                     guide, row_idx, row_const, quirk_col0, axes, unif, seed, rng_offset, index_base, mean, out, angle_out, axis_out, n);
  return check_launch();
}

int so3x_igso3_logprob_score(so3x_stream_t s, const float* R, const float* eps, int64_t eps_stride, float* logp,
                             float* score_vec, float* grad_R, int64_t n) {
  if (n < 0 || (n && (!R || !eps || !logp)) || (eps_stride != 0 && eps_stride != 1)) return SO3X_ERR_INVALID_ARG;
  if (n == 0) return SO3X_OK;
  // one 64-sample tile per wave, as many workgroups as that takes: measured against a persistent one-wave-of-blocks grid
  // with software prefetch (1261 blocks at 2^24: 175.8 us) the plain oversubscribed launch wins (65,536 blocks: 164.8 us
  // = 5.7 TB/s; 2^20: 12.1-12.3 us vs 12.5-12.7) -- the dispatcher rebalances the tail, five resident blocks per CU
  // already cover the load latency.  The in-kernel tile loop remains for n beyond the cap.
  const int64_t nt64 = (n + kWave - 1) / kWave;
  int64_t want = (nt64 + 3) / 4;
  if (want > (1 << 20)) want = 1 << 20;
  if (grad_R)
    hipLaunchKernelGGL(k_logprob_score<true>, dim3((unsigned)want), dim3(kBlock), 0, (hipStream_t)s, R, eps, eps_stride, logp, score_vec, grad_R, n);
  else
    hipLaunchKernelGGL(k_logprob_score<false>, dim3((unsigned)want), dim3(kBlock), 0, (hipStream_t)s, R, eps, eps_stride, logp, score_vec, grad_R, n);
  return check_launch();
}

}  // extern "C"
