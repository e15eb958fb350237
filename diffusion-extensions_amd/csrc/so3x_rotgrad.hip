// so3x_rotgrad.hip -- the rotation-matrix head and the autograd of the rotation ops:
//   six2rmat (util.py:67-76) forward / backward, backward of log_rmat (util.py:164-175) and rmat_dist (util.py:315-322),
//   and the fused loss_type = "prevstep" objective of SO3Diffusion.p_losses (diffusion.py:358-365).
// All HBM-streaming, one lane per sample, AoS tiles staged through LDS (so3x_common.hpp).  The reference differentiates
// these ops with torch autograd; here the gradients are the closed forms (checked against the reference's autograd
// results in tests/golden/prevstep.npz).
#include "so3x_common.hpp"
#include "so3x_math.hpp"
#include "so3x_reverse_step.hpp"

using namespace so3x;

namespace {

#define SO3X_TILE_LOOP(n)                                                         \
  const int64_t ntiles = ((n) + kTile - 1) / kTile;                               \
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x)

#define SO3X_TILE_VARS(n)                                                         \
  const int64_t base = tile * kTile;                                              \
  const int cnt = (int)(((n) - base) < kTile ? ((n) - base) : kTile);             \
  const int64_t idx = base + threadIdx.x;                                         \
  const bool live = threadIdx.x < cnt;                                            \
  (void)idx; (void)live;

struct Gram { float b1[3], b2[3], n1, n2, d; };  // the Gram-Schmidt intermediates of one sample

__device__ __forceinline__ Gram gram(const float* x) {
  Gram g;
  g.n1 = sqrtf(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
#pragma unroll
  for (int k = 0; k < 3; k++) g.b1[k] = x[k] / g.n1;
  g.d = g.b1[0] * x[3] + g.b1[1] * x[4] + g.b1[2] * x[5];
  float u[3];
#pragma unroll
  for (int k = 0; k < 3; k++) u[k] = x[3 + k] - g.d * g.b1[k];
  g.n2 = sqrtf(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
#pragma unroll
  for (int k = 0; k < 3; k++) g.b2[k] = u[k] / g.n2;
  return g;
}
__device__ __forceinline__ void cross3(const float* a, const float* b, float* o) {
  o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0];
}
__device__ __forceinline__ void six2rmat_one(const float* x, float* R) {
  const Gram g = gram(x);
#pragma unroll
  for (int k = 0; k < 3; k++) { R[k] = g.b1[k]; R[3 + k] = g.b2[k]; }
  cross3(g.b1, g.b2, R + 6);  // rows b1, b2, b1 x b2 (util.py:74-75)
}
// G = dL/dR (rows g1, g2, g3) -> dL/dx
__device__ __forceinline__ void six2rmat_bwd_one(const float* x, const float* G, float* dx) {
  const Gram g = gram(x);
  float c1[3], c2[3], gb1[3], gb2[3], du[3];
  cross3(g.b2, G + 6, c1);  // b3 = b1 x b2:  dL/db1 += b2 x g3
  cross3(G + 6, g.b1, c2);  //                dL/db2 += g3 x b1
#pragma unroll
  for (int k = 0; k < 3; k++) { gb1[k] = G[k] + c1[k]; gb2[k] = G[3 + k] + c2[k]; }
  const float p2 = gb2[0] * g.b2[0] + gb2[1] * g.b2[1] + gb2[2] * g.b2[2];
#pragma unroll
  for (int k = 0; k < 3; k++) du[k] = (gb2[k] - p2 * g.b2[k]) / g.n2;  // b2 = u / |u|
  const float q = g.b1[0] * du[0] + g.b1[1] * du[1] + g.b1[2] * du[2];
#pragma unroll
  for (int k = 0; k < 3; k++) {  // u = a2 - (b1 . a2) b1
    dx[3 + k] = du[k] - q * g.b1[k];
    gb1[k] += -g.d * du[k] - q * x[3 + k];
  }
  const float p1 = gb1[0] * g.b1[0] + gb1[1] * g.b1[1] + gb1[2] * g.b1[2];
#pragma unroll
  for (int k = 0; k < 3; k++) dx[k] = (gb1[k] - p1 * g.b1[k]) / g.n1;    // b1 = a1 / |a1|
}

// omega(M) = atan2(s, c) of util.py:165-169 and its gradient  [c/(4s) (M - M^T) - (s/2) I] / (s^2 + c^2)
// (SURVEY.md 8a row A3); at s == 0 the antisymmetric term is 0/0 and is dropped (zero gradient at the identity).
__device__ __forceinline__ float domega(const float* M, float* g, float* s_out) {
  const float v0 = M[7] - M[5], v1 = M[2] - M[6], v2 = M[3] - M[1];
  const float s = sqrtf(v0 * v0 + v1 * v1 + v2 * v2) * 0.5f;
  const float c = (M[0] + M[4] + M[8] - 1.0f) * 0.5f;
  const float iden = 1.0f / (s * s + c * c);
  const float k = s > 0.0f ? c / (4.0f * s) : 0.0f;
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) g[3 * i + j] = (k * (M[3 * i + j] - M[3 * j + i]) - (i == j ? 0.5f * s : 0.0f)) * iden;
  *s_out = s;
  return atan2_pos(s, c);
}

__global__ void __launch_bounds__(kBlock) k_six2rmat(const float* __restrict__ x, float* __restrict__ R, int64_t n) {
  __shared__ __attribute__((aligned(16))) float sm[kTile * 9];
  SO3X_TILE_LOOP(n) {
    SO3X_TILE_VARS(n)
    float v[6], r[9];
    load_rows<6>(x, base, cnt, sm, v);
    six2rmat_one(v, r);
    store_rows<9>(R, base, cnt, sm, r);
  }
}

__global__ void __launch_bounds__(kBlock)
k_six2rmat_bwd(const float* __restrict__ x, const float* __restrict__ G, float* __restrict__ dx, int64_t n) {
  __shared__ __attribute__((aligned(16))) float sm[kTile * 9];
  SO3X_TILE_LOOP(n) {
    SO3X_TILE_VARS(n)
    float v[6], g[9], d[6];
    load_rows<6>(x, base, cnt, sm, v);
    load_rows<9>(G, base, cnt, sm, g);
    six2rmat_bwd_one(v, g, d);
    store_rows<6>(dx, base, cnt, sm, d);
  }
}

// log = scale (R - R^T), scale = omega / (2 s):  dL/dR = scale (G - G^T) + <G, R - R^T> d scale/dR,
//   d scale = d omega / (2 s) - omega / (2 s^2) ds,  ds/dR = (R - R^T) / (4 s)
__global__ void __launch_bounds__(kBlock)
k_log_rmat_bwd(const float* __restrict__ R, const float* __restrict__ G, float* __restrict__ dR, int64_t n) {
  __shared__ __attribute__((aligned(16))) float sm[kTile * 9];
  SO3X_TILE_LOOP(n) {
    SO3X_TILE_VARS(n)
    float r[9], g[9], dom[9], o[9], s;
    load_rows<9>(R, base, cnt, sm, r);
    load_rows<9>(G, base, cnt, sm, g);
    const float om = domega(r, dom, &s);
    const float i2s = s > 0.0f ? 1.0f / (2.0f * s) : 0.0f;
    const float scale = om == 0.0f ? 0.0f : om * i2s;  // util.py:174
    float gs = 0.0f;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
      for (int j = 0; j < 3; j++) gs += g[3 * i + j] * (r[3 * i + j] - r[3 * j + i]);
    const float kS = om * i2s * i2s * i2s;  // omega / (2 s^2) * 1 / (4 s) = omega / (8 s^3)
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
      for (int j = 0; j < 3; j++)
        o[3 * i + j] = scale * (g[3 * i + j] - g[3 * j + i]) + gs * (dom[3 * i + j] * i2s - kS * (r[3 * i + j] - r[3 * j + i]));
    store_rows<9>(dR, base, cnt, sm, o);
  }
}

// dist = sqrt(2) omega(a^T b):  dL/da = b dM^T,  dL/db = a dM,  dM = sqrt(2) g d omega/dM
__global__ void __launch_bounds__(kBlock)
k_rmat_dist_bwd(const float* __restrict__ A, const float* __restrict__ Bm, const float* __restrict__ gd,
                float* __restrict__ dA, float* __restrict__ dB, int64_t n) {
  __shared__ __attribute__((aligned(16))) float sm[kTile * 9];
  SO3X_TILE_LOOP(n) {
    SO3X_TILE_VARS(n)
    float a[9], b[9], M[9], dom[9], o[9], s;
    load_rows<9>(A, base, cnt, sm, a);
    load_rows<9>(Bm, base, cnt, sm, b);
    mul33_at(a, b, M);
    domega(M, dom, &s);
    const float k = 1.41421356237309505f * (live ? gd[idx] : 0.0f);
#pragma unroll
    for (int i = 0; i < 9; i++) dom[i] *= k;
    mul33_bt(b, dom, o);
    store_rows<9>(dA, base, cnt, sm, o);
    mul33(a, dom, o);
    store_rows<9>(dB, base, cnt, sm, o);
  }
}

// "prevstep" objective: step = x_noisy^T (so3_scale(x_start, c1_t) so3_scale(x_noisy, c2_t)) (diffusion.py:299-302, 360-364),
// dist2 = rmat_dist(x_recon, step)^2 = 2 omega(x_recon^T step)^2; per-block double partial sums of dist2, and
// dx = gscale / n * d dist2 / d x_recon = gscale / n * step (4 omega d omega/dM)^T.
// SIX: x_recon is given as the network's raw 6 outputs; six2rmat and its backward are applied here (dx = d loss / d out6).
template <bool SIX>
__global__ void __launch_bounds__(kBlock)
k_prevstep(const float* __restrict__ sched, int T, const float* __restrict__ xr, const float* __restrict__ xs,
           const float* __restrict__ xn, const int64_t* __restrict__ t, int64_t t_stride, int64_t n, float gk,
           double* __restrict__ partial, float* __restrict__ dx, float* __restrict__ step_out) {
  __shared__ __attribute__((aligned(16))) float sm[kTile * 9];
  __shared__ double wsum[kBlock / 64];
  double acc = 0.0;
  SO3X_TILE_LOOP(n) {
    SO3X_TILE_VARS(n)
    float r[9], x0[9], xt[9], w[3], e1[9], e2[9], pm[9], st[9], M[9], dom[9], o[9], s;
    float v6[6];
    if constexpr (SIX) {
      load_rows<6>(xr, base, cnt, sm, v6);
      six2rmat_one(v6, r);
    } else {
      load_rows<9>(xr, base, cnt, sm, r);
    }
    load_rows<9>(xs, base, cnt, sm, x0);
    load_rows<9>(xn, base, cnt, sm, xt);
    int64_t tt = live ? t[idx * t_stride] : 0;
    tt = tt < 0 ? 0 : (tt >= T ? T - 1 : tt);
    const float c1 = sched[S_COEF1 * T + tt], c2 = sched[S_COEF2 * T + tt];
    log3(x0, w);
    w[0] *= c1; w[1] *= c1; w[2] *= c1;
    exp3(w, e1);
    log3(xt, w);
    w[0] *= c2; w[1] *= c2; w[2] *= c2;
    exp3(w, e2);
    mul33(e1, e2, pm);
    mul33_at(xt, pm, st);
    mul33_at(r, st, M);
    const float om = domega(M, dom, &s);
    if (live) acc += (double)(2.0f * om * om);
    if (step_out) store_rows<9>(step_out, base, cnt, sm, st);
    if (dx) {
      const float k = 4.0f * om * gk;
#pragma unroll
      for (int i = 0; i < 9; i++) dom[i] *= k;
      mul33_bt(st, dom, o);
      if constexpr (SIX) {
        float d6[6];
        six2rmat_bwd_one(v6, o, d6);
        store_rows<6>(dx, base, cnt, sm, d6);
      } else {
        store_rows<9>(dx, base, cnt, sm, o);
      }
    }
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) acc += __shfl_down(acc, d);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

__global__ void __launch_bounds__(256) k_sum_partials_f(const double* __restrict__ partial, int np, double scale, float* __restrict__ out) {
  __shared__ double sm[256];
  double s = 0.0;
  for (int i = threadIdx.x; i < np; i += 256) s += partial[i];
  sm[threadIdx.x] = s;
  __syncthreads();
  for (int d = 128; d > 0; d >>= 1) {
    if (threadIdx.x < d) sm[threadIdx.x] += sm[threadIdx.x + d];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = (float)(sm[0] * scale);
}

inline int bad(int64_t n) { return n < 0; }
#define SO3X_LAUNCH(kern, n, s, ...)                                                        \
  do {                                                                                      \
    if ((n) == 0) return SO3X_OK;                                                           \
    const int64_t nt_ = ((n) + kTile - 1) / kTile;                                          \
    hipLaunchKernelGGL(kern, dim3(grid_for_tiles(nt_)), dim3(kBlock), 0, (hipStream_t)(s), __VA_ARGS__); \
    return check_launch();                                                                  \
  } while (0)

}  // namespace

extern "C" {

int so3x_six2rmat(so3x_stream_t s, const float* x6, float* R, int64_t n) {
  if (bad(n) || (n && (!x6 || !R))) return SO3X_ERR_INVALID_ARG;
  SO3X_LAUNCH(k_six2rmat, n, s, x6, R, n);
}
int so3x_six2rmat_bwd(so3x_stream_t s, const float* x6, const float* dR, float* dx6, int64_t n) {
  if (bad(n) || (n && (!x6 || !dR || !dx6))) return SO3X_ERR_INVALID_ARG;
  SO3X_LAUNCH(k_six2rmat_bwd, n, s, x6, dR, dx6, n);
}
int so3x_log_rmat_bwd(so3x_stream_t s, const float* R, const float* dlog, float* dR, int64_t n) {
  if (bad(n) || (n && (!R || !dlog || !dR))) return SO3X_ERR_INVALID_ARG;
  SO3X_LAUNCH(k_log_rmat_bwd, n, s, R, dlog, dR, n);
}
int so3x_rmat_dist_bwd(so3x_stream_t s, const float* a, const float* b, const float* ddist, float* da, float* db, int64_t n) {
  if (bad(n) || (n && (!a || !b || !ddist || !da || !db))) return SO3X_ERR_INVALID_ARG;
  SO3X_LAUNCH(k_rmat_dist_bwd, n, s, a, b, ddist, da, db, n);
}

size_t so3x_prevstep_workspace_bytes(int64_t n) { (void)n; return 2048 * sizeof(double); }

int so3x_prevstep_loss(so3x_stream_t s, const float* sched, int T, const float* x_recon, const float* x_start,
                       const float* x_noisy, const int64_t* t, int64_t t_stride, int64_t n, float* loss, float* dx_recon,
                       float* step_out, void* workspace, size_t workspace_bytes) {
  if (n <= 0 || T <= 0 || !sched || !x_recon || !x_start || !x_noisy || !t || !loss || (t_stride != 0 && t_stride != 1))
    return SO3X_ERR_INVALID_ARG;
  if (!workspace || workspace_bytes < so3x_prevstep_workspace_bytes(n)) return SO3X_ERR_WORKSPACE;
  const int grid = grid_for_tiles((n + kTile - 1) / kTile);
  hipLaunchKernelGGL(k_prevstep<false>, dim3(grid), dim3(kBlock), 0, (hipStream_t)s, sched, T, x_recon, x_start, x_noisy, t, t_stride, n,
                     1.0f / (float)n, reinterpret_cast<double*>(workspace), dx_recon, step_out);
  hipLaunchKernelGGL(k_sum_partials_f, dim3(1), dim3(256), 0, (hipStream_t)s, (const double*)workspace, grid, 1.0 / (double)n, loss);
  return check_launch();
}

int so3x_prevstep_loss6(so3x_stream_t s, const float* sched, int T, const float* out6, const float* x_start,
                        const float* x_noisy, const int64_t* t, int64_t t_stride, int64_t n, float* loss, float* dout6,
                        void* workspace, size_t workspace_bytes) {
  if (n <= 0 || T <= 0 || !sched || !out6 || !x_start || !x_noisy || !t || !loss || (t_stride != 0 && t_stride != 1))
    return SO3X_ERR_INVALID_ARG;
  if (!workspace || workspace_bytes < so3x_prevstep_workspace_bytes(n)) return SO3X_ERR_WORKSPACE;
  const int grid = grid_for_tiles((n + kTile - 1) / kTile);
  hipLaunchKernelGGL(k_prevstep<true>, dim3(grid), dim3(kBlock), 0, (hipStream_t)s, sched, T, out6, x_start, x_noisy, t, t_stride, n,
                     1.0f / (float)n, reinterpret_cast<double*>(workspace), dout6, (float*)nullptr);
  hipLaunchKernelGGL(k_sum_partials_f, dim3(1), dim3(256), 0, (hipStream_t)s, (const double*)workspace, grid, 1.0 / (double)n, loss);
  return check_launch();
}

}  // extern "C"
