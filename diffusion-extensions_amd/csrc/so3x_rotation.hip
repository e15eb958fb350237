// so3x_rotation.hip -- standalone rotation-algebra kernels.
// All are HBM-streaming: one lane per rotation, AoS tiles staged through LDS so every
// global access is a 16-B-per-lane coalesced transfer (so3x_common.hpp).
#include "so3x_common.hpp"
#include "so3x_math.hpp"

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

__global__ void __launch_bounds__(kBlock) k_quat_to_rmat(const float* __restrict__ q, float* __restrict__ R, int64_t n) {
  __shared__ __attribute__((aligned(16))) float sm[kTile * 9];
  SO3X_TILE_LOOP(n) {
    SO3X_TILE_VARS(n)
    __syncthreads();
    tile_to_lds<4>(q, base, cnt, sm);
    __syncthreads();
    float4 qq = reinterpret_cast<const float4*>(sm)[threadIdx.x];
    float qv[4] = {qq.x, qq.y, qq.z, qq.w}, r[9];
    quat_to_rmat(qv, r);
    store_rows<9>(R, base, cnt, sm, r);
  }
}

// mode 0: skew matrix out [n][3][3]; mode 1: vee vector out [n][3]
template <int MODE>
__global__ void __launch_bounds__(kBlock) k_log_rmat(const float* __restrict__ R, float* __restrict__ out, int64_t n) {
  __shared__ __attribute__((aligned(16))) float sm[kTile * 9];
  SO3X_TILE_LOOP(n) {
    SO3X_TILE_VARS(n)
    float r[9], w[3];
    load_rows<9>(R, base, cnt, sm, r);
    log3(r, w);
    if (MODE == 1) {
      store_rows<3>(out, base, cnt, sm, w);
    } else {
      // vec2skew, util.py:87-92: S21 = w0, S20 = -w1, S10 = w2, antisymmetric
      float s[9] = {0.f, -w[2], w[1], w[2], 0.f, -w[0], -w[1], w[0], 0.f};
      store_rows<9>(out, base, cnt, sm, s);
    }
  }
}

// util.py:95-107  orthogonalise: U round(S) V^T of the SVD M = U S V^T, evaluated as  M V diag(round(s_i) / s_i) V^T  with
// V, s^2 from a cyclic Jacobi eigen-decomposition of the symmetric M^T M (fp32, 6 sweeps; a singular value that rounds to 0
// drops its term, so no left vector is ever formed from a vanishing s).  Invariant to the choice of V inside a repeated
// singular value, and the identity map (to rounding) on a rotation.
__device__ __forceinline__ void jacobi_rot(float (&a)[3][3], float (&v)[3][3], int p, int q) {
  const float apq = a[p][q];
  if (fabsf(apq) < 1e-30f) return;
  const float tau = (a[q][q] - a[p][p]) / (2.0f * apq);
  const float t = (tau >= 0.f ? 1.0f : -1.0f) / (fabsf(tau) + sqrtf(1.0f + tau * tau));
  const float c = 1.0f / sqrtf(1.0f + t * t), sn = t * c;
  const int r = 3 - p - q;
  const float app = a[p][p], aqq = a[q][q], arp = a[r][p], arq = a[r][q];
  a[p][p] = app - t * apq;
  a[q][q] = aqq + t * apq;
  a[p][q] = a[q][p] = 0.0f;
  a[r][p] = a[p][r] = c * arp - sn * arq;
  a[r][q] = a[q][r] = sn * arp + c * arq;
#pragma unroll
  for (int k = 0; k < 3; k++) {
    const float vkp = v[k][p], vkq = v[k][q];
    v[k][p] = c * vkp - sn * vkq;
    v[k][q] = sn * vkp + c * vkq;
  }
}

__global__ void __launch_bounds__(kBlock) k_orthogonalise(const float* __restrict__ M, float* __restrict__ out, int64_t n) {
  __shared__ __attribute__((aligned(16))) float sm[kTile * 9];
  SO3X_TILE_LOOP(n) {
    SO3X_TILE_VARS(n)
    float m[9], o[9];
    load_rows<9>(M, base, cnt, sm, m);
    float a[3][3], v[3][3] = {{1.f, 0.f, 0.f}, {0.f, 1.f, 0.f}, {0.f, 0.f, 1.f}};
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
      for (int j = 0; j < 3; j++) a[i][j] = m[i] * m[j] + m[3 + i] * m[3 + j] + m[6 + i] * m[6 + j];  // (M^T M)_ij
#pragma unroll 1
    for (int sweep = 0; sweep < 6; sweep++) {
      jacobi_rot(a, v, 0, 1);
      jacobi_rot(a, v, 0, 2);
      jacobi_rot(a, v, 1, 2);
    }
    float w[3];
#pragma unroll
    for (int i = 0; i < 3; i++) {
      const float sv = sqrtf(fmaxf(a[i][i], 0.0f));
      const float r = rintf(sv);  // torch.round: half to even
      w[i] = r > 0.0f ? r / sv : 0.0f;
    }
    float g[3][3];  // V diag(w) V^T
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
      for (int j = 0; j < 3; j++) g[i][j] = v[i][0] * w[0] * v[j][0] + v[i][1] * w[1] * v[j][1] + v[i][2] * w[2] * v[j][2];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
      for (int j = 0; j < 3; j++) o[3 * i + j] = m[3 * i] * g[0][j] + m[3 * i + 1] * g[1][j] + m[3 * i + 2] * g[2][j];
    store_rows<9>(out, base, cnt, sm, o);
  }
}

__global__ void __launch_bounds__(kBlock) k_exp_skewvec(const float* __restrict__ v, float* __restrict__ R, int64_t n) {
  __shared__ __attribute__((aligned(16))) float sm[kTile * 9];
  SO3X_TILE_LOOP(n) {
    SO3X_TILE_VARS(n)
    float w[3], r[9];
    load_rows<3>(v, base, cnt, sm, w);
    exp3(w, r);
    store_rows<9>(R, base, cnt, sm, r);
  }
}

__global__ void __launch_bounds__(kBlock)
k_so3_scale(const float* __restrict__ R, const float* __restrict__ k, int64_t k_stride, float* __restrict__ out, int64_t n) {
  __shared__ __attribute__((aligned(16))) float sm[kTile * 9];
  SO3X_TILE_LOOP(n) {
    SO3X_TILE_VARS(n)
    float r[9], w[3], o[9];
    load_rows<9>(R, base, cnt, sm, r);
    float kk = live ? load_scalar(k, k_stride, idx) : 0.f;
    log3(r, w);
    w[0] *= kk; w[1] *= kk; w[2] *= kk;
    exp3(w, o);
    store_rows<9>(out, base, cnt, sm, o);
  }
}

__global__ void __launch_bounds__(kBlock)
k_aa_to_rmat(const float* __restrict__ axis, const float* __restrict__ ang, float* __restrict__ R, int64_t n) {
  __shared__ __attribute__((aligned(16))) float sm[kTile * 9];
  SO3X_TILE_LOOP(n) {
    SO3X_TILE_VARS(n)
    float a[3], r[9];
    load_rows<3>(axis, base, cnt, sm, a);
    float th = live ? ang[idx] : 0.f;
    float nrm = sqrtf(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]);  // util.py:201
    float w[3] = {a[0] / nrm * th, a[1] / nrm * th, a[2] / nrm * th};
    exp3(w, r);  // matrix_exp (204); orthogonalise (205) is the identity map on a rotation
    store_rows<9>(R, base, cnt, sm, r);
  }
}

__global__ void __launch_bounds__(kBlock)
k_rmat_to_aa(const float* __restrict__ R, float* __restrict__ axis, float* __restrict__ ang, int64_t n) {
  __shared__ __attribute__((aligned(16))) float sm[kTile * 9];
  SO3X_TILE_LOOP(n) {
    SO3X_TILE_VARS(n)
    float r[9], w[3];
    load_rows<9>(R, base, cnt, sm, r);
    log3(r, w);
    float a = sqrtf(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
    float ax[3] = {w[0] / a, w[1] / a, w[2] / a};  // NaN at a == 0: reference behaviour (util.py:218)
    if (live) ang[idx] = a;
    store_rows<3>(axis, base, cnt, sm, ax);
  }
}

__global__ void __launch_bounds__(kBlock)
k_so3_lerp(const float* __restrict__ A, int64_t a_stride, const float* __restrict__ Bm, const float* __restrict__ wgt,
           int64_t w_stride, float* __restrict__ out, int64_t n) {
  __shared__ __attribute__((aligned(16))) float sm[kTile * 9];
  SO3X_TILE_LOOP(n) {
    SO3X_TILE_VARS(n)
    float a[9], b[9], c[9], w[3], rc[9], o[9];
    if (a_stride == 0) {
#pragma unroll
      for (int j = 0; j < 9; j++) a[j] = A[j];
    } else {
      load_rows<9>(A, base, cnt, sm, a);
    }
    load_rows<9>(Bm, base, cnt, sm, b);
    float wt = live ? load_scalar(wgt, w_stride, idx) : 0.f;
    mul33_at(a, b, c);                                              // util.py:333
    log3(c, w);                                                     // rmat_to_aa (334)
    float ang = sqrtf(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
    float ax[3] = {w[0] / ang, w[1] / ang, w[2] / ang};             // NaN at 0, as the reference
    float nrm = sqrtf(ax[0] * ax[0] + ax[1] * ax[1] + ax[2] * ax[2]);  // aa_to_rmat renormalises (201)
    float ia = wt * ang;                                            // 336
    float wi[3] = {ax[0] / nrm * ia, ax[1] / nrm * ia, ax[2] / nrm * ia};
    exp3(wi, rc);                                                   // 337
    mul33(a, rc, o);                                                // 338
    store_rows<9>(out, base, cnt, sm, o);
  }
}

__global__ void __launch_bounds__(kBlock)
k_rmat_dist(const float* __restrict__ A, const float* __restrict__ Bm, float* __restrict__ out, int64_t n) {
  __shared__ __attribute__((aligned(16))) float sm[kTile * 9];
  SO3X_TILE_LOOP(n) {
    SO3X_TILE_VARS(n)
    float a[9], b[9], c[9], w[3];
    load_rows<9>(A, base, cnt, sm, a);
    load_rows<9>(Bm, base, cnt, sm, b);
    mul33_at(a, b, c);
    log3(c, w);
    if (live) out[idx] = sqrtf(2.0f * (w[0] * w[0] + w[1] * w[1] + w[2] * w[2]));  // Frobenius norm of the skew log
  }
}

__global__ void __launch_bounds__(kBlock)
k_rmul(const float* __restrict__ A, int64_t a_stride, const float* __restrict__ Bm, int64_t b_stride, int transpose_b,
       float* __restrict__ out, int64_t n) {
  __shared__ __attribute__((aligned(16))) float sm[kTile * 9];
  SO3X_TILE_LOOP(n) {
    SO3X_TILE_VARS(n)
    float a[9], b[9], o[9];
    if (a_stride == 0) {
#pragma unroll
      for (int j = 0; j < 9; j++) a[j] = A[j];
    } else {
      load_rows<9>(A, base, cnt, sm, a);
    }
    if (b_stride == 0) {
#pragma unroll
      for (int j = 0; j < 9; j++) b[j] = Bm[j];
    } else {
      load_rows<9>(Bm, base, cnt, sm, b);
    }
    if (transpose_b) mul33_bt(a, b, o); else mul33(a, b, o);
    store_rows<9>(out, base, cnt, sm, o);
  }
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

int so3x_quat_to_rmat(so3x_stream_t s, const float* q, float* R, int64_t n) {
  if (bad(n) || (n && (!q || !R))) return SO3X_ERR_INVALID_ARG;
  SO3X_LAUNCH(k_quat_to_rmat, n, s, q, R, n);
}
int so3x_log_rmat(so3x_stream_t s, const float* R, float* log_out, int64_t n) {
  if (bad(n) || (n && (!R || !log_out))) return SO3X_ERR_INVALID_ARG;
  SO3X_LAUNCH(k_log_rmat<0>, n, s, R, log_out, n);
}
int so3x_log_rmat_vec(so3x_stream_t s, const float* R, float* vec_out, int64_t n) {
  if (bad(n) || (n && (!R || !vec_out))) return SO3X_ERR_INVALID_ARG;
  SO3X_LAUNCH(k_log_rmat<1>, n, s, R, vec_out, n);
}
int so3x_orthogonalise(so3x_stream_t s, const float* M, float* out, int64_t n) {
  if (bad(n) || (n && (!M || !out))) return SO3X_ERR_INVALID_ARG;
  SO3X_LAUNCH(k_orthogonalise, n, s, M, out, n);
}
int so3x_exp_skewvec(so3x_stream_t s, const float* v, float* R, int64_t n) {
  if (bad(n) || (n && (!v || !R))) return SO3X_ERR_INVALID_ARG;
  SO3X_LAUNCH(k_exp_skewvec, n, s, v, R, n);
}
int so3x_so3_scale(so3x_stream_t s, const float* R, const float* k, int64_t k_stride, float* out, int64_t n) {
  if (bad(n) || (n && (!R || !k || !out)) || (k_stride != 0 && k_stride != 1)) return SO3X_ERR_INVALID_ARG;
  SO3X_LAUNCH(k_so3_scale, n, s, R, k, k_stride, out, n);
}
int so3x_aa_to_rmat(so3x_stream_t s, const float* axis, const float* angle, float* R, int64_t n) {
  if (bad(n) || (n && (!axis || !angle || !R))) return SO3X_ERR_INVALID_ARG;
  SO3X_LAUNCH(k_aa_to_rmat, n, s, axis, angle, R, n);
}
int so3x_rmat_to_aa(so3x_stream_t s, const float* R, float* axis, float* angle, int64_t n) {
  if (bad(n) || (n && (!R || !axis || !angle))) return SO3X_ERR_INVALID_ARG;
  SO3X_LAUNCH(k_rmat_to_aa, n, s, R, axis, angle, n);
}
int so3x_so3_lerp(so3x_stream_t s, const float* a, int64_t a_stride, const float* b, const float* w, int64_t w_stride,
                  float* out, int64_t n) {
  if (bad(n) || (n && (!a || !b || !w || !out)) || (a_stride != 0 && a_stride != 9) || (w_stride != 0 && w_stride != 1))
    return SO3X_ERR_INVALID_ARG;
  SO3X_LAUNCH(k_so3_lerp, n, s, a, a_stride, b, w, w_stride, out, n);
}
int so3x_rmat_dist(so3x_stream_t s, const float* a, const float* b, float* out, int64_t n) {
  if (bad(n) || (n && (!a || !b || !out))) return SO3X_ERR_INVALID_ARG;
  SO3X_LAUNCH(k_rmat_dist, n, s, a, b, out, n);
}
int so3x_rmul(so3x_stream_t s, const float* a, int64_t a_stride, const float* b, int64_t b_stride, int transpose_b,
              float* out, int64_t n) {
  if (bad(n) || (n && (!a || !b || !out)) || (a_stride != 0 && a_stride != 9) || (b_stride != 0 && b_stride != 9))
    return SO3X_ERR_INVALID_ARG;
  SO3X_LAUNCH(k_rmul, n, s, a, a_stride, b, b_stride, transpose_b, out, n);
}

}  // extern "C"
