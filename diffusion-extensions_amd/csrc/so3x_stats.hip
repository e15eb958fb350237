// so3x_stats.hip -- sample-quality statistics: the O(N^2) kernel sums behind the
// reference's MMD / kernel two-sample test (util.py:110-134, 254-312).  One launch replaces the reference's
// chunked [N,N,3,3] broadcast (bingham_test.py:29 needs chunksize 4000 to fit 20,000^2 pairs in memory).
#include "so3x_common.hpp"
#include "so3x_math.hpp"

using namespace so3x;

namespace {

constexpr int YT = 256;  // y rotations staged per LDS tile (9 KB)

// kernel value of one pair from the entries the reference's formulas touch:
//   KIND 0: rmat_gaussian_kernel = exp(-rmat_dist) = exp(-sqrt(2) * angle(x^T y))     (util.py:128-134, 315-322)
//   KIND 1: rmat_cosine_kernel   = (tr(y^T x) - 1) / 2                                 (util.py:136-151)
template <int KIND>
__device__ __forceinline__ float pair_kernel(const float* x, const float* y) {
  const float tr = x[0] * y[0] + x[1] * y[1] + x[2] * y[2] + x[3] * y[3] + x[4] * y[4] + x[5] * y[5] + x[6] * y[6] +
                   x[7] * y[7] + x[8] * y[8];
  const float c = (tr - 1.0f) * 0.5f;
  if (KIND == 1) return c;
  // M = x^T y;  vee(M - M^T) = (M21 - M12, M02 - M20, M10 - M01),  M[a][b] = sum_k x[k][a] y[k][b]
  const float v0 = (x[2] * y[1] + x[5] * y[4] + x[8] * y[7]) - (x[1] * y[2] + x[4] * y[5] + x[7] * y[8]);
  const float v1 = (x[0] * y[2] + x[3] * y[5] + x[6] * y[8]) - (x[2] * y[0] + x[5] * y[3] + x[8] * y[6]);
  const float v2 = (x[1] * y[0] + x[4] * y[3] + x[7] * y[6]) - (x[0] * y[1] + x[3] * y[4] + x[6] * y[7]);
  const float s = fsqrt(v0 * v0 + v1 * v1 + v2 * v2) * 0.5f;
  return __expf(-1.41421356237309505f * atan2_pos(s, c));
}

// grid = (x tiles, y splits).  Each lane keeps one x in registers and sweeps its split of Y through LDS.
template <int KIND>
__global__ void __launch_bounds__(kBlock)
k_pair_sum(const float* __restrict__ X, int64_t nx, const float* __restrict__ Y, int64_t ny, double* __restrict__ partial) {
  __shared__ __attribute__((aligned(16))) float sm[YT * 9];
  __shared__ double wsum[kBlock / 64];
  const int64_t xi = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  const bool live = xi < nx;
  float x[9];
  load_rot9(X, live ? xi : nx - 1, x);
  const int64_t ytiles = (ny + YT - 1) / YT;
  double acc = 0.0;
  for (int64_t yt = blockIdx.y; yt < ytiles; yt += gridDim.y) {
    const int64_t yb = yt * YT;
    const int cnt = (int)((ny - yb) < YT ? (ny - yb) : YT);
    __syncthreads();
    tile_to_lds<9>(Y, yb, cnt, sm);
    __syncthreads();
    float part = 0.0f;
#pragma unroll 4
    for (int j = 0; j < cnt; j++) part += pair_kernel<KIND>(x, sm + 9 * j);  // same LDS address in every lane: broadcast
    acc += (double)part;
  }
  if (!live) acc = 0.0;
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) acc += __shfl_down(acc, d);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int w = 0; w < kBlock / 64; w++) t += wsum[w];
    partial[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = t;
  }
}

__global__ void __launch_bounds__(256) k_sum_partials(const double* __restrict__ partial, int n, double scale, float* __restrict__ out) {
  __shared__ double sh[256];
  double t = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) t += partial[i];  // fixed order: deterministic
  sh[threadIdx.x] = t;
  __syncthreads();
  for (int d = 128; d > 0; d >>= 1) {
    if (threadIdx.x < d) sh[threadIdx.x] += sh[threadIdx.x + d];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = (float)(sh[0] * scale);
}

}  // namespace

extern "C" {

size_t so3x_kernel_sum_workspace_bytes(int64_t nx, int64_t ny) {
  (void)ny;
  const int64_t gx = (nx + kBlock - 1) / kBlock;
  return (size_t)(gx > 0 ? gx : 1) * 64 * sizeof(double);
}

int so3x_kernel_sum(so3x_stream_t s, const float* X, int64_t nx, const float* Y, int64_t ny, int kind, float scale,
                    float* out, void* workspace, size_t workspace_bytes) {
  if (nx < 0 || ny < 0 || !out || ((nx && ny) && (!X || !Y)) || (kind != 0 && kind != 1)) return SO3X_ERR_INVALID_ARG;
  if (!workspace || workspace_bytes < so3x_kernel_sum_workspace_bytes(nx, ny)) return SO3X_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)s;
  if (nx == 0 || ny == 0) {
    hipError_t e = hipMemsetAsync(out, 0, sizeof(float), st);
    return e == hipSuccess ? SO3X_OK : (int)e;
  }
  const int64_t gx = (nx + kBlock - 1) / kBlock;
  const int64_t ytiles = (ny + YT - 1) / YT;
  int64_t gy = (2048 + gx - 1) / gx;  // ~2048 blocks in flight
  if (gy > ytiles) gy = ytiles;
  if (gy > 64) gy = 64;
  if (gy < 1) gy = 1;
  if (gx > 0x7fffffff) return SO3X_ERR_INVALID_ARG;
  double* partial = reinterpret_cast<double*>(workspace);
  if (kind == 0) hipLaunchKernelGGL(k_pair_sum<0>, dim3((unsigned)gx, (unsigned)gy), dim3(kBlock), 0, st, X, nx, Y, ny, partial);
  else hipLaunchKernelGGL(k_pair_sum<1>, dim3((unsigned)gx, (unsigned)gy), dim3(kBlock), 0, st, X, nx, Y, ny, partial);
  hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(256), 0, st, (const double*)partial, (int)(gx * gy), (double)scale, out);
  return check_launch();
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------
// F.mse_loss of p_losses (diffusion.py:357) and its gradient: loss = mean((a - b)^2) over all elements.
// Two-stage deterministic reduction (per-block double partials, fixed-order final sum).
// ---------------------------------------------------------------------------------------------
namespace {

__global__ void __launch_bounds__(256) k_mse_partial(const float* __restrict__ a, const float* __restrict__ b, int64_t n,
                                                     double* __restrict__ partial) {
  __shared__ double wsum[4];
  double acc = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float d = a[i] - b[i];
    acc += (double)(d * d);
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) acc += __shfl_down(acc, d);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// grad_a = (a - b) * 2/n * gscale[0]   (gscale = the upstream gradient of the scalar loss, device-resident: no sync)
__global__ void __launch_bounds__(256) k_mse_grad(const float* __restrict__ a, const float* __restrict__ b, int64_t n,
                                                  const float* __restrict__ gscale, float* __restrict__ ga) {
  const float k = 2.0f / (float)n * (gscale ? gscale[0] : 1.0f);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) ga[i] = (a[i] - b[i]) * k;
}

}  // namespace

extern "C" {

size_t so3x_mse_workspace_bytes(int64_t n) { (void)n; return 1024 * sizeof(double); }

int so3x_mse_loss(so3x_stream_t s, const float* a, const float* b, int64_t n, float* loss, void* workspace, size_t workspace_bytes) {
  if (n <= 0 || !a || !b || !loss) return SO3X_ERR_INVALID_ARG;
  if (!workspace || workspace_bytes < so3x_mse_workspace_bytes(n)) return SO3X_ERR_WORKSPACE;
  const int64_t want = (n + 255) / 256;
  const int grid = (int)(want < 1024 ? want : 1024);
  hipLaunchKernelGGL(k_mse_partial, dim3(grid), dim3(256), 0, (hipStream_t)s, a, b, n, reinterpret_cast<double*>(workspace));
  hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(256), 0, (hipStream_t)s, (const double*)workspace, grid, 1.0 / (double)n, loss);
  return check_launch();
}

int so3x_mse_grad(so3x_stream_t s, const float* a, const float* b, int64_t n, const float* gscale, float* grad_a) {
  if (n <= 0 || !a || !b || !grad_a) return SO3X_ERR_INVALID_ARG;
  const int64_t want = (n + 255) / 256;
  hipLaunchKernelGGL(k_mse_grad, dim3((unsigned)(want < 2048 ? want : 2048)), dim3(256), 0, (hipStream_t)s, a, b, n, gscale, grad_a);
  return check_launch();
}

}  // extern "C"
