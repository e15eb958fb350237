// so3x_planenet.hip -- the PlaneNet point-cloud denoiser (reference models.py:185-210: SIREN position encoding || sinusoidal time
// embedding -> `layers` x nn.TransformerEncoderLayer(dim, heads) [post-norm, ReLU, feed-forward 2048; eval mode, or training mode with its dropout] -> PoolRN ->
// Linear(dim, 3)), forward and backward, as hand-written kernels.
//
// This file: the EXACT-FP32 form (SO3X_PREC_F32) for any (dim, heads, layers, ffn) and the host-side plan both precisions share.
// Every product runs on v_mfma_f32_32x32x2_f32 (bit-for-bit a k-ordered fp32 fmaf chain) through ONE strided, batched GEMM
// kernel; attention materialises its [B][H][P][P] probabilities (they are also what the backward needs).  The bf16 form of the
// aircraft task's own shape (dim 512, 4 heads) lives in so3x_planenet_bf16.hip: tiled bf16 GEMMs with fused epilogues and a
// flash-style attention that never writes the scores.
//
// Activations are token-major [B*P][width] (the reference runs nn.TransformerEncoder sequence-first, [P][B][dim]; the
// arithmetic per token is the same).
#include "so3x_planenet.hpp"
#include "so3x_math.hpp"

namespace so3x {
namespace plane {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// ------------------------------------------------------------------------------------------------ strided batched GEMM
// C[z][m][n] = act(alpha * sum_k A[z](m, k) B[z](k, n) + bias[n] (+ C[z][m][n] if accumulate)),  z = (z0, z1), z1 < nb1.
// Operands by element strides (any of row-major / transposed / a slice of a wider matrix), C rows contiguous in n.
struct GemmP {
  const float* A;
  const float* B;
  float* C;
  const float* bias;
  int M, N, K;
  int64_t sam, sak, sbk, sbn, scm;
  int nb1;
  int64_t sA0, sA1, sB0, sB1, sC0, sC1;
  float alpha;
  int relu, accumulate;
  int kc;   // > 0: split-K -- batch index z0 is the K-chunk [z0 kc, min(K, (z0 + 1) kc)); the chunks' partial products go to C + z0 sC0
};

constexpr int GB = 64, GK = 16, GPAD = 68;

__global__ __launch_bounds__(256) void k_gemm_f32(const GemmP p) {
  __shared__ float As[GK][GPAD];
  __shared__ float Bs[GK][GPAD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.y * GB, n0 = blockIdx.x * GB;
  const int z0 = blockIdx.z / p.nb1, z1 = blockIdx.z % p.nb1;
  const float* A = p.A + z0 * p.sA0 + z1 * p.sA1;
  const float* B = p.B + z0 * p.sB0 + z1 * p.sB1;
  float* C = p.C + z0 * p.sC0 + z1 * p.sC1;
  const int Kn = p.kc > 0 ? (p.K - z0 * p.kc < p.kc ? p.K - z0 * p.kc : p.kc) : p.K;   // this workgroup's extent of K
  const bool a_kfast = p.sak == 1, b_nfast = p.sbn == 1;
  int am[4], ak[4], bn[4], bk[4];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int e = tid + 256 * i;
    if (a_kfast) { ak[i] = e & 15; am[i] = e >> 4; } else { am[i] = e & 63; ak[i] = e >> 6; }
    if (b_nfast) { bn[i] = e & 63; bk[i] = e >> 6; } else { bk[i] = e & 15; bn[i] = e >> 4; }
  }
  float ra[4], rb[4];
  auto fetch = [&](int k0) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int gm = m0 + am[i], gk = k0 + ak[i];
      ra[i] = (gm < p.M && gk < Kn) ? A[gm * p.sam + gk * p.sak] : 0.f;
      const int gn = n0 + bn[i], gk2 = k0 + bk[i];
      rb[i] = (gn < p.N && gk2 < Kn) ? B[gk2 * p.sbk + gn * p.sbn] : 0.f;
    }
  };
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; r++) acc[r] = 0.f;
  fetch(0);
  for (int k0 = 0; k0 < Kn; k0 += GK) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
      As[ak[i]][am[i]] = ra[i];
      Bs[bk[i]][bn[i]] = rb[i];
    }
    __syncthreads();
    if (k0 + GK < Kn) fetch(k0 + GK);
#pragma unroll
    for (int kk = 0; kk < GK / 2; kk++) {
      const float a = As[2 * kk + (lane >> 5)][wm * 32 + (lane & 31)];
      const float b = Bs[2 * kk + (lane >> 5)][wn * 32 + (lane & 31)];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    __syncthreads();
  }
  const int col = n0 + wn * 32 + (lane & 31);
  if (col < p.N) {
    const float bv = p.bias ? p.bias[col] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const int row = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      if (row < p.M) {
        float v = p.alpha * acc[r] + bv;
        float* c = C + (int64_t)row * p.scm + col;
        if (p.accumulate) v += *c;
        if (p.relu) v = v > 0.f ? v : 0.f;
        *c = v;
      }
    }
  }
}

// C = alpha A B (+ bias) ..., A: M x K, B: K x N  (declared in so3x_planenet.hpp)
int gemm(hipStream_t s, Mat A, Mat B, float* C, int64_t ldc, int M, int N, int K, const float* bias, float alpha, bool relu, bool accumulate,
         int nb0, int nb1, int64_t sA0, int64_t sA1, int64_t sB0, int64_t sB1, int64_t sC0, int64_t sC1) {
  if (M <= 0 || N <= 0 || nb0 * nb1 <= 0) return SO3X_OK;
  GemmP p{A.p, B.p, C, bias, M, N, K, A.s0, A.s1, B.s0, B.s1, ldc, nb1, sA0, sA1, sB0, sB1, sC0, sC1, alpha, relu ? 1 : 0, accumulate ? 1 : 0, 0};
  const int gy = (M + GB - 1) / GB;
  if (gy > 65535 || nb0 * nb1 > 65535) return SO3X_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(k_gemm_f32, dim3((N + GB - 1) / GB, gy, nb0 * nb1), dim3(256), 0, s, p);
  return check_launch();
}

// C = A B for a SMALL M x N and a LONG K (a weight gradient: K = every token of the batch; at the reference's own batch sizes the
// 2048-long products of the feed-forward block as well): as gemm() the launch would be a handful of workgroups walking all of K (dW of
// a 64-wide layer: ONE workgroup).  Here K is cut into chunks, one workgroup per (tile, chunk) writes its partial product to
// slab[chunk][M][N], and k_splitk_sum adds the chunks in a fixed order (deterministic, no atomics), then the bias / the old C.
// 32 outputs per workgroup, eight groups of threads each walking every eighth chunk (four loads in flight), combined in a fixed order
// (one thread per output walking all chunks took 122 us for a 64 x 64 gradient cut 792 ways)
__global__ __launch_bounds__(256) void k_splitk_sum(const float* __restrict__ slab, int nchunk, int64_t mn, int N, float* __restrict__ C, int64_t ldc,
                                                    const float* __restrict__ bias, int accumulate) {
  __shared__ float red[8][32];
  const int64_t i = (int64_t)blockIdx.x * 32 + (threadIdx.x & 31);
  const int g = threadIdx.x >> 5;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  if (i < mn) {
    int z = g;
    for (; z + 24 < nchunk; z += 32) {
      a0 += slab[(int64_t)z * mn + i];
      a1 += slab[(int64_t)(z + 8) * mn + i];
      a2 += slab[(int64_t)(z + 16) * mn + i];
      a3 += slab[(int64_t)(z + 24) * mn + i];
    }
    for (; z < nchunk; z += 8) a0 += slab[(int64_t)z * mn + i];
  }
  red[g][threadIdx.x & 31] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (g == 0 && i < mn) {
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < 8; k++) acc += red[k][threadIdx.x];
    float* c = C + (i / N) * ldc + i % N;
    if (bias) acc += bias[i % N];
    if (accumulate) acc += *c;
    *c = acc;
  }
}
int gemm_splitk(hipStream_t s, Mat A, Mat B, float* C, int64_t ldc, int M, int N, int K, float* slab, size_t slab_floats, const float* bias,
                bool accumulate) {
  if (M <= 0 || N <= 0) return SO3X_OK;
  const int64_t tiles = (int64_t)((M + GB - 1) / GB) * ((N + GB - 1) / GB), mn = (int64_t)M * N;
  int nch = (int)(1024 / tiles);                                   // ~4 workgroups per CU
  // ... of at least 512 of K each where that still fills the chip; down to 128 where it does not: at 1,584 tokens (prot_train.py's
  // --batch 4) 512 made three chunks -- three workgroups of 33 sequential K tiles (27 us) for a 64 x 64 gradient
  const int kmin = tiles * (K / 512) >= 256 ? 512 : 128;
  if ((int64_t)nch * kmin > K) nch = K / kmin;
  if ((size_t)nch * (size_t)mn > slab_floats) nch = (int)(slab_floats / (size_t)mn);
  if (nch < 2) return gemm(s, A, B, C, ldc, M, N, K, bias, 1.f, false, accumulate);
  const int kc = ((K + nch - 1) / nch + GK - 1) / GK * GK;
  nch = (K + kc - 1) / kc;
  GemmP p{A.p, B.p, slab, nullptr, M, N, K, A.s0, A.s1, B.s0, B.s1, N, 1, (int64_t)kc * A.s1, 0, (int64_t)kc * B.s0, 0, mn, 0, 1.f, 0, 0, kc};
  const int gy = (M + GB - 1) / GB;
  if (gy > 65535 || nch > 65535) return SO3X_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(k_gemm_f32, dim3((N + GB - 1) / GB, gy, nch), dim3(256), 0, s, p);
  hipLaunchKernelGGL(k_splitk_sum, dim3((unsigned)((mn + 31) / 32)), dim3(256), 0, s, slab, nch, mn, N, C, ldc, bias, accumulate ? 1 : 0);
  return check_launch();
}

// ------------------------------------------------------------------------------------------------ pointwise / row kernels
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}

// SIREN position encoding + sinusoidal time embedding (models.py:50-72, 13-25, 199-201):
//   pre[n][j] = Wp[j] . x[n] + bp[j],  s = sin(pre)            (the post_scale Linear is a GEMM on s, written to h0[:, :d2])
//   h0[n][d2 + j] = j < d2/2 ? sin(t_b f_j) : cos(t_b f_{j - d2/2}),  f_j = exp(j * -(ln 1e4 / (d2/2 - 1)))   (fp32, as torch)
__global__ __launch_bounds__(256) void k_embed(const float* __restrict__ x, const int64_t* __restrict__ t, const float* __restrict__ wp,
                                               const float* __restrict__ bp, float* __restrict__ pre, float* __restrict__ sn,
                                               float* __restrict__ h0, int64_t N, int64_t P, int d2, int d, float neg_emb) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= N * d2) return;
  const int64_t n = idx / d2;
  const int j = (int)(idx - n * d2);
  const float* xp = x + n * 3;
  float a = bp[j];
  a = fmaf(xp[0], wp[j * 3 + 0], a);
  a = fmaf(xp[1], wp[j * 3 + 1], a);
  a = fmaf(xp[2], wp[j * 3 + 2], a);
  if (pre) pre[idx] = a;
  sn[idx] = sinf(a);
  const int half = d2 / 2;
  const int jj = j < half ? j : j - half;
  const float f = (float)exp((double)((float)jj * neg_emb));
  const float arg = (float)t[n / P] * f;
  h0[n * d + d2 + j] = j < half ? sinf(arg) : cosf(arg);
}

// rows of S -> softmax in place (torch.softmax(dim=-1)); one wave per row
__global__ __launch_bounds__(256) void k_softmax_rows(float* __restrict__ S, int64_t rows, int cols) {
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  float* r = S + row * cols;
  float m = -INFINITY;
  for (int c = lane; c < cols; c += 64) m = fmaxf(m, r[c]);
  m = wave_max(m);
  float sum = 0.f;
  for (int c = lane; c < cols; c += 64) {
    const float e = expf(r[c] - m);
    r[c] = e;
    sum += e;
  }
  sum = wave_sum(sum);
  const float inv = 1.f / sum;
  for (int c = lane; c < cols; c += 64) r[c] *= inv;
}

// dS = P o (dP - rowsum(dP o P)) * scale, in place over dP
__global__ __launch_bounds__(256) void k_softmax_bwd(const float* __restrict__ Pm, float* __restrict__ dP, int64_t rows, int cols, float scale) {
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  const float* p = Pm + row * cols;
  float* g = dP + row * cols;
  float s = 0.f;
  if (cols <= 512) {   // both rows in registers: read once (the same operations in the same order as below)
    float pv[8], gv[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const int c = lane + 64 * j;
      pv[j] = c < cols ? p[c] : 0.f;
      gv[j] = c < cols ? g[c] : 0.f;
      if (c < cols) s = fmaf(pv[j], gv[j], s);
    }
    s = wave_sum(s);
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const int c = lane + 64 * j;
      if (c < cols) g[c] = pv[j] * (gv[j] - s) * scale;
    }
    return;
  }
  for (int c = lane; c < cols; c += 64) s = fmaf(p[c], g[c], s);
  s = wave_sum(s);
  for (int c = lane; c < cols; c += 64) g[c] = p[c] * (g[c] - s) * scale;
}

// r = a + b; y = LayerNorm(r) * gamma + beta (biased variance, eps inside the root: torch.nn.LayerNorm); stats[n] = (mean, rstd)
__global__ __launch_bounds__(256) void k_add_ln(const float* __restrict__ a, const float* b, float* r_out,   // r_out may be b
                                                float* __restrict__ y, float* __restrict__ stats, const float* __restrict__ gamma,
                                                const float* __restrict__ beta, int64_t N, int d, float eps) {
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= N) return;
  const int lane = threadIdx.x & 63;
  const float* ap = a + row * d;
  const float* bp = b ? b + row * d : nullptr;
  float s = 0.f;
  for (int c = lane; c < d; c += 64) s += ap[c] + (bp ? bp[c] : 0.f);
  const float mean = wave_sum(s) / d;
  float v = 0.f;
  for (int c = lane; c < d; c += 64) {
    const float u = ap[c] + (bp ? bp[c] : 0.f) - mean;
    v = fmaf(u, u, v);
  }
  const float rstd = 1.f / sqrtf(wave_sum(v) / d + eps);
  for (int c = lane; c < d; c += 64) {
    const float u = ap[c] + (bp ? bp[c] : 0.f);
    if (r_out) r_out[row * d + c] = u;
    y[row * d + c] = (u - mean) * rstd * gamma[c] + beta[c];
  }
  if (stats && lane == 0) {
    stats[row * 2] = mean;
    stats[row * 2 + 1] = rstd;
  }
}

// dr = rstd (dy gamma - mean(dy gamma) - xhat mean(dy gamma xhat)),  xhat = (r - mean) rstd
__global__ __launch_bounds__(256) void k_ln_bwd(const float* __restrict__ dy, const float* __restrict__ r, const float* __restrict__ stats,
                                                const float* __restrict__ gamma, float* __restrict__ dr, int64_t N, int d) {
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= N) return;
  const int lane = threadIdx.x & 63;
  const float mean = stats[row * 2], rstd = stats[row * 2 + 1];
  float c1 = 0.f, c2 = 0.f;
  for (int c = lane; c < d; c += 64) {
    const float g = dy[row * d + c] * gamma[c], xh = (r[row * d + c] - mean) * rstd;
    c1 += g;
    c2 = fmaf(g, xh, c2);
  }
  c1 = wave_sum(c1) / d;
  c2 = wave_sum(c2) / d;
  for (int c = lane; c < d; c += 64) {
    const float g = dy[row * d + c] * gamma[c], xh = (r[row * d + c] - mean) * rstd;
    dr[row * d + c] = rstd * (g - c1 - xh * c2);
  }
}

// column sums in a fixed order: part[chunk][c] = sum over the chunk's rows of X[row][c] (* xhat[row][c] when r/stats are given:
// LayerNorm's d gamma), then k_colsum_final adds the chunks.  ch rows per chunk (colsum_ch).
__global__ __launch_bounds__(256) void k_colsum_part(const float* __restrict__ X, int64_t ld, int64_t rows, int cols, const float* __restrict__ r,
                                                     int64_t ldr, const float* __restrict__ stats, float* __restrict__ part, int ch) {
  __shared__ float red[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), g = threadIdx.x >> 6;
  const int64_t r0 = (int64_t)blockIdx.y * ch, r1 = r0 + ch < rows ? r0 + ch : rows;
  float acc = 0.f;
  if (c < cols) {
    for (int64_t i = r0 + g; i < r1; i += 4) {
      float v = X[i * ld + c];
      if (r) v *= (r[i * ldr + c] - stats[i * 2]) * stats[i * 2 + 1];
      acc += v;
    }
  }
  red[g][threadIdx.x & 63] = acc;
  __syncthreads();
  if (g == 0 && c < cols) part[(int64_t)blockIdx.y * cols + c] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}
// 32 columns per workgroup, eight groups of threads each walking every eighth chunk (four loads in flight), combined in a fixed order
__global__ __launch_bounds__(256) void k_colsum_final(const float* __restrict__ part, int nchunks, int cols, float* __restrict__ out) {
  __shared__ float red[8][32];
  const int c = blockIdx.x * 32 + (threadIdx.x & 31), g = threadIdx.x >> 5;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  if (c < cols) {
    int i = g;
    for (; i + 24 < nchunks; i += 32) {
      a0 += part[(int64_t)i * cols + c];
      a1 += part[(int64_t)(i + 8) * cols + c];
      a2 += part[(int64_t)(i + 16) * cols + c];
      a3 += part[(int64_t)(i + 24) * cols + c];
    }
    for (; i < nchunks; i += 8) a0 += part[(int64_t)i * cols + c];
  }
  red[g][threadIdx.x & 31] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (g == 0 && c < cols) {
    float a = 0.f;
#pragma unroll
    for (int k = 0; k < 8; k++) a += red[k][threadIdx.x];
    out[c] = a;
  }
}

__global__ __launch_bounds__(256) void k_relu_bwd(float* __restrict__ df, const float* __restrict__ f, int64_t n, float scale) {
  const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;   // four elements per thread
  if (i + 3 < n && (((uintptr_t)df | (uintptr_t)f) & 15) == 0) {
    float4 d = *reinterpret_cast<float4*>(df + i);
    const float4 v = *reinterpret_cast<const float4*>(f + i);
    d.x = v.x > 0.f ? d.x * scale : 0.f; d.y = v.y > 0.f ? d.y * scale : 0.f;   // (scale = 1 / keep when the hidden activations were dropped out)
    d.z = v.z > 0.f ? d.z * scale : 0.f; d.w = v.w > 0.f ? d.w * scale : 0.f;
    *reinterpret_cast<float4*>(df + i) = d;
    return;
  }
  for (int64_t j = i; j < n && j < i + 4; j++) df[j] = f[j] > 0.f ? df[j] * scale : 0.f;
}
// dst[e] = keep(e) ? src[e] / (1 - p) : 0 over a flat array (src may be dst); eight elements per thread, one Philox call (Drop)
__global__ __launch_bounds__(256) void k_dropout(const float* src, float* dst, int64_t n, uint32_t thr16, float inv_keep, uint64_t seed,
                                                 uint64_t ctr_hi) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (8 * i >= n) return;
  const Philox4 r = philox4x32_10(seed, (uint64_t)i, ctr_hi);
  const uint32_t u[4] = {r.x, r.y, r.z, r.w};
  if (8 * i + 7 < n && (((uintptr_t)src | (uintptr_t)dst) & 15) == 0) {
    // the thread's eight elements as two 16-byte accesses each way (element by element -- eight loads and eight stores 32 bytes apart
    // across the lanes -- the pass ran at a third of the rate: 7.5 of ProtNet's 30.7 ms per training evaluation at 256 complexes)
    const float4 a = reinterpret_cast<const float4*>(src + 8 * i)[0], b = reinterpret_cast<const float4*>(src + 8 * i)[1];
    auto keep = [&](int j, float v) { return ((u[j >> 1] >> (16 * (j & 1))) & 0xFFFFu) >= thr16 ? v * inv_keep : 0.f; };
    reinterpret_cast<float4*>(dst + 8 * i)[0] = float4{keep(0, a.x), keep(1, a.y), keep(2, a.z), keep(3, a.w)};
    reinterpret_cast<float4*>(dst + 8 * i)[1] = float4{keep(4, b.x), keep(5, b.y), keep(6, b.z), keep(7, b.w)};
    return;
  }
#pragma unroll
  for (int j = 0; j < 8; j++)
    if (8 * i + j < n) dst[8 * i + j] = ((u[j >> 1] >> (16 * (j & 1))) & 0xFFFFu) >= thr16 ? src[8 * i + j] * inv_keep : 0.f;
}
__global__ __launch_bounds__(256) void k_cos_mul(float* __restrict__ ds, const float* __restrict__ pre, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) ds[i] *= cosf(pre[i]);
}

// PoolRN (models.py:94-110), all-true mask: w_p = sigmoid(x_p . wpool + bpool); xs_b = sum_p w_p x_p / max(sum_p w_p, 1e-6).
// (lin is linear, so sum_p w_p lin(x_p) / S = lin(xs): the [P][dim] x [dim][dim] product collapses to one row per cloud.)
__global__ __launch_bounds__(256) void k_pool_logits(const float* __restrict__ x, const float* __restrict__ wpool, const float* __restrict__ bpool,
                                                     float* __restrict__ w, int64_t N, int d) {
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= N) return;
  const int lane = threadIdx.x & 63;
  float s = 0.f;
  for (int c = lane; c < d; c += 64) s = fmaf(x[row * d + c], wpool[c], s);
  s = wave_sum(s) + bpool[0];
  if (lane == 0) w[row] = 1.f / (1.f + expf(-s));
}
__global__ __launch_bounds__(256) void k_pool_sum(const float* __restrict__ x, const float* __restrict__ w, float* __restrict__ S,
                                                  float* __restrict__ xs, int64_t P, int d) {
  __shared__ float red[4][64];
  __shared__ float sred[4];
  const int b = blockIdx.y, c = blockIdx.x * 64 + (threadIdx.x & 63), g = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const float* xb = x + (int64_t)b * P * d;
  const float* wb = w + (int64_t)b * P;
  float acc = 0.f, sw = 0.f;
  for (int64_t p = g; p < P; p += 4) {
    const float wv = wb[p];
    sw += wv;
    if (c < d) acc = fmaf(wv, xb[p * d + c], acc);
  }
  red[g][lane] = acc;
  if (lane == 0) sred[g] = sw;
  __syncthreads();
  if (g == 0) {
    const float tot = (sred[0] + sred[1]) + (sred[2] + sred[3]);
    const float v = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
    if (c < d) xs[(int64_t)b * d + c] = v / fmaxf(tot, 1e-6f);
    if (blockIdx.x == 0 && lane == 0) S[b] = tot;
  }
}
// token-level backward of the pooling: g_p = d logit_p, dx_p = (w_p / S) dxs + g_p wpool
__global__ __launch_bounds__(256) void k_pool_bwd_tok(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ S,
                                                      const float* __restrict__ xs, const float* __restrict__ dxs, const float* __restrict__ wpool,
                                                      float* __restrict__ dx, float* __restrict__ g, int64_t N, int64_t P, int d) {
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= N) return;
  const int lane = threadIdx.x & 63;
  const int64_t b = row / P;
  const float* dxb = dxs + b * d;
  float a = 0.f, c0 = 0.f;
  for (int c = lane; c < d; c += 64) {
    a = fmaf(x[row * d + c], dxb[c], a);
    c0 = fmaf(xs[b * d + c], dxb[c], c0);
  }
  a = wave_sum(a);
  c0 = wave_sum(c0);
  const float Sv = S[b], Sc = fmaxf(Sv, 1e-6f), wv = w[row];
  const float e = (a - (Sv >= 1e-6f ? c0 : 0.f)) / Sc;   // d / d w_p (the clamp passes no gradient to S below 1e-6)
  const float gv = e * wv * (1.f - wv);
  for (int c = lane; c < d; c += 64) dx[row * d + c] = (wv / Sc) * dxb[c] + gv * wpool[c];
  if (lane == 0) g[row] = gv;
}

// PoolRN's lin and the output Linear (models.py:100-109, 195-196): pooled = Wlin xs + blin -- one wave per output row, 32 rows per
// workgroup, the row read as contiguous 256-byte pieces -- then out = Wout pooled + bout, one workgroup per cloud.  Exact fp32,
// fixed summation order.
__global__ __launch_bounds__(256) void k_head_lin(const float* __restrict__ xs, const float* __restrict__ wlin, const float* __restrict__ blin,
                                                  float* __restrict__ pooled, int d) {
  const int b = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float* xb = xs + (int64_t)b * d;
  const int j0 = blockIdx.x * 32 + wave * 8;
  float a[8];
#pragma unroll
  for (int r = 0; r < 8; r++) a[r] = 0.f;
  for (int c = lane; c < d; c += 64) {
    const float xv = xb[c];
#pragma unroll
    for (int r = 0; r < 8; r++) a[r] = fmaf(wlin[(int64_t)(j0 + r < d ? j0 + r : d - 1) * d + c], xv, a[r]);
  }
#pragma unroll
  for (int r = 0; r < 8; r++) {
    const float v = wave_sum(a[r]);
    if (lane == 0 && j0 + r < d) pooled[(int64_t)b * d + j0 + r] = v + blin[j0 + r];
  }
}
__global__ __launch_bounds__(256) void k_head_out(const float* __restrict__ pooled, const float* __restrict__ wout, const float* __restrict__ bout,
                                                  float* __restrict__ out, int d) {
  const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (wave >= 3) return;
  float a = 0.f;
  for (int c = lane; c < d; c += 64) a = fmaf(wout[(int64_t)wave * d + c], pooled[(int64_t)b * d + c], a);
  a = wave_sum(a) + bout[wave];
  if (lane == 0) out[(int64_t)b * 3 + wave] = a;
}
int head(hipStream_t s, const float* xs, const float* wlin, const float* blin, const float* wout, const float* bout, float* pooled, float* out,
         int64_t B, int d) {
  hipLaunchKernelGGL(k_head_lin, dim3((d + 31) / 32, (unsigned)B), dim3(256), 0, s, xs, wlin, blin, pooled, d);
  hipLaunchKernelGGL(k_head_out, dim3((unsigned)B), dim3(256), 0, s, pooled, wout, bout, out, d);
  return check_launch();
}

inline unsigned blocks_for(int64_t n, int per) { return (unsigned)((n + per - 1) / per); }

// ------------------------------------------------------------------------------------------------ buffers
struct LayerActs { float *qkv, *probs, *o, *r1, *st1, *x1, *f, *r2, *st2; };
struct Acts {
  float *pre, *sn;
  float* h[66];            // h[l] = input of layer l; h[L] = the encoder's output
  LayerActs layer[65];
  float *w, *S, *xs, *pooled;
  size_t bytes;
};
// per_layer: every layer keeps its own buffers (what the backward reads); else the layers share one set and h ping-pongs
inline Acts carve_acts(const Shape& s, void* mem, bool per_layer) {
  Acts a;
  Carve c(mem);
  const size_t N = (size_t)s.N(), d = s.d;
  a.pre = per_layer ? c.take<float>(N * s.d2()) : nullptr;
  a.sn = c.take<float>(N * s.d2());
  if (per_layer) {
    for (int l = 0; l <= s.L; l++) a.h[l] = c.take<float>(N * d);
  } else {
    float* h0 = c.take<float>(N * d);
    float* h1 = c.take<float>(N * d);
    for (int l = 0; l <= s.L; l++) a.h[l] = (l & 1) ? h1 : h0;
  }
  for (int l = 0; l < s.L; l++) {
    if (l == 0 || per_layer) {
      LayerActs& k = a.layer[l];
      k.qkv = c.take<float>(N * 3 * d);
      k.probs = c.take<float>((size_t)s.B * s.H * s.P * s.P);
      k.o = c.take<float>(N * d);
      k.r1 = c.take<float>(N * d);
      k.st1 = c.take<float>(N * 2);
      k.x1 = c.take<float>(N * d);
      k.f = c.take<float>(N * s.F);
      k.r2 = c.take<float>(N * d);
      k.st2 = c.take<float>(N * 2);
    } else {
      a.layer[l] = a.layer[0];
    }
  }
  a.w = c.take<float>(N);
  a.S = c.take<float>((size_t)s.B);
  a.xs = c.take<float>((size_t)s.B * d);
  a.pooled = c.take<float>((size_t)s.B * d);
  a.bytes = c.off;
  return a;
}

struct BwdBufs {
  float *dA, *dB, *dF, *dqkv, *dO, *dprobs, *ds, *dpooled, *dxs, *g, *part;
  size_t bytes;
};
inline BwdBufs carve_bwd(const Shape& s, void* mem) {
  BwdBufs b;
  Carve c(mem);
  const size_t N = (size_t)s.N(), d = s.d;
  b.dA = c.take<float>(N * d);
  b.dB = c.take<float>(N * d);
  b.dF = c.take<float>(N * s.F);
  b.dqkv = c.take<float>(N * 3 * d);
  b.dO = c.take<float>(N * d);
  b.dprobs = c.take<float>((size_t)s.B * s.H * s.P * s.P);
  b.ds = c.take<float>(N * s.d2());
  b.dpooled = c.take<float>((size_t)s.B * d);
  b.dxs = c.take<float>((size_t)s.B * d);
  b.g = c.take<float>(N);
  const size_t widest = (size_t)(s.F > 3 * s.d ? s.F : 3 * s.d);
  b.part = c.take<float>(colsum_part_floats(s.N(), widest));
  b.bytes = c.off;
  return b;
}

int colsum(hipStream_t s, const float* X, int64_t ld, int64_t rows, int cols, float* out, float* part, const float* r, int64_t ldr,
           const float* stats) {
  const int ch = colsum_ch(rows, cols), nch = (int)((rows + ch - 1) / ch);
  hipLaunchKernelGGL(k_colsum_part, dim3((cols + 63) / 64, nch), dim3(256), 0, s, X, ld, rows, cols, r, ldr, stats, part, ch);
  hipLaunchKernelGGL(k_colsum_final, dim3((cols + 31) / 32), dim3(256), 0, s, part, nch, cols, out);
  return check_launch();
}

// host-side entry points of the row kernels for the other translation units (so3x_protnet.hip)
int dropout_apply(hipStream_t s, const Drop& dr, int layer, int site, const float* src, float* dst, int64_t n) {
  hipLaunchKernelGGL(k_dropout, dim3(blocks_for((n + 7) / 8, 256)), dim3(256), 0, s, src, dst, n, dr.thr16(), dr.inv_keep(), dr.seed, dr.ctr_hi(layer, site));
  return check_launch();
}
int add_ln(hipStream_t s, const float* a, const float* b, float* r_out, float* y, float* stats, const float* gamma, const float* beta, int64_t N,
           int d, float eps) {
  hipLaunchKernelGGL(k_add_ln, dim3(blocks_for(N, 4)), dim3(256), 0, s, a, b, r_out, y, stats, gamma, beta, N, d, eps);
  return check_launch();
}
int ln_bwd(hipStream_t s, const float* dy, const float* r, const float* stats, const float* gamma, float* dr, int64_t N, int d) {
  hipLaunchKernelGGL(k_ln_bwd, dim3(blocks_for(N, 4)), dim3(256), 0, s, dy, r, stats, gamma, dr, N, d);
  return check_launch();
}
int softmax_bwd(hipStream_t s, const float* probs, float* dprobs, int64_t rows, int cols, float scale) {
  hipLaunchKernelGGL(k_softmax_bwd, dim3(blocks_for(rows, 4)), dim3(256), 0, s, probs, dprobs, rows, cols, scale);
  return check_launch();
}
int relu_bwd(hipStream_t s, float* df, const float* f, int64_t n, float scale) {
  hipLaunchKernelGGL(k_relu_bwd, dim3(blocks_for((n + 3) / 4, 256)), dim3(256), 0, s, df, f, n, scale);
  return check_launch();
}
int cos_mul(hipStream_t s, float* ds, const float* pre, int64_t n) {
  hipLaunchKernelGGL(k_cos_mul, dim3(blocks_for(n, 256)), dim3(256), 0, s, ds, pre, n);
  return check_launch();
}

#define TRY(expr)                 \
  do {                            \
    int rc__ = (expr);            \
    if (rc__) return rc__;        \
  } while (0)

// ------------------------------------------------------------------------------------------------ forward plan (fp32)
static int dropout(hipStream_t s, const Drop& dr, int layer, int site, const float* src, float* dst, int64_t n) {
  hipLaunchKernelGGL(k_dropout, dim3(blocks_for((n + 7) / 8, 256)), dim3(256), 0, s, src, dst, n, dr.thr16(), dr.inv_keep(), dr.seed, dr.ctr_hi(layer, site));
  return check_launch();
}

// dr.on(): a training-mode forward; `pd` = room for one layer's dropped-out probabilities ([B][H][P][P]; the stash keeps the
// plain softmax, which its backward needs everywhere)
int forward_f32(hipStream_t s, const Shape& sh, const float* prm, const float* x, const int64_t* t, float* out, float* encoding_out,
                const Acts& a, const Drop& dr, float* pd) {
  const ParamOff po = param_offsets(sh);
  const int64_t N = sh.N();
  const int d = sh.d, d2 = sh.d2(), H = sh.H, dh = sh.dh(), F = sh.F;
  const int64_t P = sh.P;
  const int half = d2 / 2;
  const float neg_emb = (float)(-(log(10000.0) / (half - 1)));
  hipLaunchKernelGGL(k_embed, dim3(blocks_for(N * d2, 256)), dim3(256), 0, s, x, t, prm + po.wp, prm + po.bp, a.pre, a.sn, a.h[0], N, P, d2, d, neg_emb);
  TRY(check_launch());
  TRY(gemm(s, rowmajor(a.sn, d2), transposed(prm + po.wps, d2), a.h[0], d, (int)N, d2, d2, prm + po.bps));   // post_scale -> h0[:, :d2]
  const float scale = 1.f / sqrtf((float)dh);
  for (int l = 0; l < sh.L; l++) {
    const LayerOff lo = po.layer(l);
    const LayerActs& k = a.layer[l];
    const float* h = a.h[l];
    TRY(gemm(s, rowmajor(h, d), transposed(prm + lo.wqkv, d), k.qkv, 3 * d, (int)N, 3 * d, d, prm + lo.bqkv));
    // scores[b][h] = (Q K^T) / sqrt(dh), softmax over keys, O = P V
    TRY(gemm(s, rowmajor(k.qkv, 3 * d), transposed(k.qkv + d, 3 * d), k.probs, P, (int)P, (int)P, dh, nullptr, scale, false, false, (int)sh.B, H,
             P * 3 * d, dh, P * 3 * d, dh, (int64_t)H * P * P, P * P));
    hipLaunchKernelGGL(k_softmax_rows, dim3(blocks_for(sh.B * H * P, 4)), dim3(256), 0, s, k.probs, sh.B * H * P, (int)P);
    TRY(check_launch());
    const float* pv = k.probs;
    if (dr.on()) {
      TRY(dropout(s, dr, l, DROP_ATTN, k.probs, pd, sh.B * H * P * P));
      pv = pd;
    }
    TRY(gemm(s, rowmajor(pv, P), rowmajor(k.qkv + 2 * d, 3 * d), k.o, d, (int)P, dh, (int)P, nullptr, 1.f, false, false, (int)sh.B, H,
             (int64_t)H * P * P, P * P, P * 3 * d, dh, P * d, dh));
    TRY(gemm(s, rowmajor(k.o, d), transposed(prm + lo.wo, d), k.r1, d, (int)N, d, d, prm + lo.bo));
    if (dr.on()) TRY(dropout(s, dr, l, DROP_BLOCK1, k.r1, k.r1, N * d));
    hipLaunchKernelGGL(k_add_ln, dim3(blocks_for(N, 4)), dim3(256), 0, s, h, k.r1, k.r1, k.x1, k.st1, prm + lo.g1, prm + lo.be1, N, d, 1e-5f);
    TRY(check_launch());
    TRY(gemm(s, rowmajor(k.x1, d), transposed(prm + lo.w1, d), k.f, F, (int)N, F, d, prm + lo.b1, 1.f, true));
    if (dr.on()) TRY(dropout(s, dr, l, DROP_FFN, k.f, k.f, N * F));   // (the stash holds the dropped-out activations: what linear2 saw)
    TRY(gemm(s, rowmajor(k.f, F), transposed(prm + lo.w2, F), k.r2, d, (int)N, d, F, prm + lo.b2));
    if (dr.on()) TRY(dropout(s, dr, l, DROP_BLOCK2, k.r2, k.r2, N * d));
    hipLaunchKernelGGL(k_add_ln, dim3(blocks_for(N, 4)), dim3(256), 0, s, k.x1, k.r2, k.r2, a.h[l + 1], k.st2, prm + lo.g2, prm + lo.be2, N, d, 1e-5f);
    TRY(check_launch());
  }
  const float* enc = a.h[sh.L];
  if (encoding_out) {
    hipError_t e = hipMemcpyAsync(encoding_out, enc, (size_t)N * d * sizeof(float), hipMemcpyDeviceToDevice, s);
    if (e != hipSuccess) return (int)e;
  }
  hipLaunchKernelGGL(k_pool_logits, dim3(blocks_for(N, 4)), dim3(256), 0, s, enc, prm + po.wpool, prm + po.bpool, a.w, N, d);
  hipLaunchKernelGGL(k_pool_sum, dim3((d + 63) / 64, (unsigned)sh.B), dim3(256), 0, s, enc, a.w, a.S, a.xs, P, d);
  TRY(check_launch());
  TRY(head(s, a.xs, prm + po.wlin, prm + po.blin, prm + po.wout, prm + po.bout, a.pooled, out, sh.B, d));
  return SO3X_OK;
}

// ------------------------------------------------------------------------------------------------ backward plan (fp32)
// dparams (overwritten) = d sum(out * dout) / d params, from the forward's per-layer buffers
int backward_f32(hipStream_t s, const Shape& sh, const float* prm, const float* x, const float* dout, float* dprm, const Acts& a,
                 const BwdBufs& w, const Drop& dr) {
  const ParamOff po = param_offsets(sh);
  const int64_t N = sh.N(), P = sh.P;
  const int d = sh.d, d2 = sh.d2(), H = sh.H, dh = sh.dh(), F = sh.F, Bn = (int)sh.B;
  const float* enc = a.h[sh.L];
  // head: out = pooled Wout^T + bout, pooled = xs Wlin^T + blin
  TRY(gemm(s, transposed(dout, 3), rowmajor(a.pooled, d), dprm + po.wout, d, 3, d, Bn));
  TRY(colsum(s, dout, 3, Bn, 3, dprm + po.bout, w.part));
  TRY(gemm(s, rowmajor(dout, 3), rowmajor(prm + po.wout, d), w.dpooled, d, Bn, d, 3));
  TRY(gemm(s, transposed(w.dpooled, d), rowmajor(a.xs, d), dprm + po.wlin, d, d, d, Bn));
  TRY(colsum(s, w.dpooled, d, Bn, d, dprm + po.blin, w.part));
  TRY(gemm(s, rowmajor(w.dpooled, d), rowmajor(prm + po.wlin, d), w.dxs, d, Bn, d, d));
  hipLaunchKernelGGL(k_pool_bwd_tok, dim3(blocks_for(N, 4)), dim3(256), 0, s, enc, a.w, a.S, a.xs, w.dxs, prm + po.wpool, w.dA, w.g, N, P, d);
  TRY(check_launch());
  TRY(gemm(s, Mat{w.g, 0, 1}, rowmajor(enc, d), dprm + po.wpool, d, 1, d, (int)N));   // sum_p g_p x_p
  TRY(colsum(s, w.g, 1, N, 1, dprm + po.bpool, w.part));
  const float scale = 1.f / sqrtf((float)dh);
  float* dcur = w.dA;    // gradient with respect to the current layer's output
  float* dalt = w.dB;
  for (int l = sh.L - 1; l >= 0; l--) {
    const LayerOff lo = po.layer(l);
    const LayerActs& k = a.layer[l];
    const float* h = a.h[l];
    // norm2 over r2 = x1 + ffn(x1)
    TRY(colsum(s, dcur, d, N, d, dprm + lo.g2, w.part, k.r2, d, k.st2));
    TRY(colsum(s, dcur, d, N, d, dprm + lo.be2, w.part));
    hipLaunchKernelGGL(k_ln_bwd, dim3(blocks_for(N, 4)), dim3(256), 0, s, dcur, k.r2, k.st2, prm + lo.g2, dalt, N, d);
    TRY(check_launch());
    // dalt = d r2: flows to x1 (residual) and through [dropout] linear2 / [dropout] relu / linear1
    const float* dy2 = dalt;
    if (dr.on()) {   // the feed-forward branch sees the gradient through its output dropout (w.dO is idle until the attention block)
      TRY(dropout(s, dr, l, DROP_BLOCK2, dalt, w.dO, N * d));
      dy2 = w.dO;
    }
    TRY(gemm(s, transposed(dy2, d), rowmajor(k.f, F), dprm + lo.w2, F, d, F, (int)N));
    TRY(colsum(s, dy2, d, N, d, dprm + lo.b2, w.part));
    TRY(gemm(s, rowmajor(dy2, d), rowmajor(prm + lo.w2, F), w.dF, F, (int)N, F, d));
    hipLaunchKernelGGL(k_relu_bwd, dim3(blocks_for((N * F + 3) / 4, 256)), dim3(256), 0, s, w.dF, k.f, N * F, dr.on() ? dr.inv_keep() : 1.f);
    TRY(check_launch());
    TRY(gemm(s, transposed(w.dF, F), rowmajor(k.x1, d), dprm + lo.w1, d, F, d, (int)N));
    TRY(colsum(s, w.dF, F, N, F, dprm + lo.b1, w.part));
    TRY(gemm(s, rowmajor(w.dF, F), rowmajor(prm + lo.w1, d), dalt, d, (int)N, d, F, nullptr, 1.f, false, true));   // dalt = d x1
    // norm1 over r1 = h + attn(h)
    TRY(colsum(s, dalt, d, N, d, dprm + lo.g1, w.part, k.r1, d, k.st1));
    TRY(colsum(s, dalt, d, N, d, dprm + lo.be1, w.part));
    hipLaunchKernelGGL(k_ln_bwd, dim3(blocks_for(N, 4)), dim3(256), 0, s, dalt, k.r1, k.st1, prm + lo.g1, dcur, N, d);
    TRY(check_launch());
    // dcur = d r1: flows to h (residual) and through [dropout] out_proj / attention / in_proj
    const float* dy1 = dcur;
    if (dr.on()) {   // (w.dF is idle from here on)
      TRY(dropout(s, dr, l, DROP_BLOCK1, dcur, w.dF, N * d));
      dy1 = w.dF;
    }
    TRY(gemm(s, transposed(dy1, d), rowmajor(k.o, d), dprm + lo.wo, d, d, d, (int)N));
    TRY(colsum(s, dy1, d, N, d, dprm + lo.bo, w.part));
    TRY(gemm(s, rowmajor(dy1, d), rowmajor(prm + lo.wo, d), w.dO, d, (int)N, d, d));
    const int64_t sq = P * 3 * d, sp = (int64_t)H * P * P;
    // dV = P^T dO (the dropped-out probabilities the forward multiplied V with, rebuilt in the buffer dP takes next)
    const float* pv = k.probs;
    if (dr.on()) {
      TRY(dropout(s, dr, l, DROP_ATTN, k.probs, w.dprobs, sh.B * H * P * P));
      pv = w.dprobs;
    }
    TRY(gemm(s, transposed(pv, P), rowmajor(w.dO, d), w.dqkv + 2 * d, 3 * d, (int)P, dh, (int)P, nullptr, 1.f, false, false, Bn, H, sp, P * P,
             P * d, dh, sq, dh));
    // dP = dO V^T [through the dropout], dS = softmax'(P, dP) / sqrt(dh)
    TRY(gemm(s, rowmajor(w.dO, d), transposed(k.qkv + 2 * d, 3 * d), w.dprobs, P, (int)P, (int)P, dh, nullptr, 1.f, false, false, Bn, H, P * d, dh,
             sq, dh, sp, P * P));
    if (dr.on()) TRY(dropout(s, dr, l, DROP_ATTN, w.dprobs, w.dprobs, sh.B * H * P * P));
    hipLaunchKernelGGL(k_softmax_bwd, dim3(blocks_for(sh.B * H * P, 4)), dim3(256), 0, s, k.probs, w.dprobs, sh.B * H * P, (int)P, scale);
    TRY(check_launch());
    // dQ = dS K, dK = dS^T Q
    TRY(gemm(s, rowmajor(w.dprobs, P), rowmajor(k.qkv + d, 3 * d), w.dqkv, 3 * d, (int)P, dh, (int)P, nullptr, 1.f, false, false, Bn, H, sp, P * P,
             sq, dh, sq, dh));
    TRY(gemm(s, transposed(w.dprobs, P), rowmajor(k.qkv, 3 * d), w.dqkv + d, 3 * d, (int)P, dh, (int)P, nullptr, 1.f, false, false, Bn, H, sp, P * P,
             sq, dh, sq, dh));
    TRY(gemm(s, transposed(w.dqkv, 3 * d), rowmajor(h, d), dprm + lo.wqkv, d, 3 * d, d, (int)N));
    TRY(colsum(s, w.dqkv, 3 * d, N, 3 * d, dprm + lo.bqkv, w.part));
    TRY(gemm(s, rowmajor(w.dqkv, 3 * d), rowmajor(prm + lo.wqkv, d), dcur, d, (int)N, d, 3 * d, nullptr, 1.f, false, true));   // dcur = d h
  }
  // embedding: h0[:, :d2] = sin(pre) Wps^T + bps, pre = x Wp^T + bp (the time-embedding half has no parameters)
  TRY(gemm(s, transposed(dcur, d), rowmajor(a.sn, d2), dprm + po.wps, d2, d2, d2, (int)N));
  TRY(colsum(s, dcur, d, N, d2, dprm + po.bps, w.part));
  TRY(gemm(s, rowmajor(dcur, d), rowmajor(prm + po.wps, d2), w.ds, d2, (int)N, d2, d2));
  hipLaunchKernelGGL(k_cos_mul, dim3(blocks_for(N * d2, 256)), dim3(256), 0, s, w.ds, a.pre, N * d2);
  TRY(check_launch());
  TRY(gemm(s, transposed(w.ds, d2), rowmajor(x, 3), dprm + po.wp, 3, d2, 3, (int)N));
  TRY(colsum(s, w.ds, d2, N, d2, dprm + po.bp, w.part));
  return SO3X_OK;
}

}  // namespace plane
}  // namespace so3x

// ------------------------------------------------------------------------------------------------ C ABI
using namespace so3x::plane;

namespace so3x { namespace plane {
// so3x_planenet_bf16.hip
bool bf16_supported(const Shape& sh);
size_t bf16_weights_bytes(const Shape& sh);
size_t bf16_workspace_bytes(const Shape& sh);
size_t bf16_stash_bytes(const Shape& sh);
int forward_bf16(hipStream_t s, const Shape& sh, const float* prm, const float* x, const int64_t* t, float* out, float* encoding_out,
                 void* stash, void* workspace, const void* prepared, const Drop& dr);
int weights_bf16(hipStream_t s, const Shape& sh, const float* prm, void* wimg, bool fold);
int backward_bf16(hipStream_t s, const Shape& sh, const float* prm, const float* x, const int64_t* t, const float* dout, float* dprm,
                  const void* stash, void* workspace, const Drop& dr);
} }

extern "C" {

int64_t so3x_planenet_param_count(int dim, int heads, int layers, int ffn) {
  Shape sh{0, 1, dim, heads, layers, ffn};
  if (!shape_ok(sh) || layers > 64) return SO3X_ERR_INVALID_ARG;
  return param_offsets(sh).total;
}

size_t so3x_planenet_stash_bytes(int64_t B, int64_t P, int dim, int heads, int layers, int ffn, int precision) {
  Shape sh{B, P, dim, heads, layers, ffn};
  if (!shape_ok(sh) || layers > 64) return 0;
  if (precision == SO3X_PREC_BF16) return bf16_supported(sh) ? bf16_stash_bytes(sh) : 0;
  return carve_acts(sh, nullptr, true).bytes;
}

size_t so3x_planenet_workspace_bytes(int64_t B, int64_t P, int dim, int heads, int layers, int ffn, int precision) {
  Shape sh{B, P, dim, heads, layers, ffn};
  if (!shape_ok(sh) || layers > 64) return 0;
  if (precision == SO3X_PREC_BF16) return bf16_supported(sh) ? bf16_workspace_bytes(sh) : 0;
  const size_t f = carve_acts(sh, nullptr, false).bytes, b = carve_bwd(sh, nullptr).bytes;
  return f > b ? f : b;
}

size_t so3x_planenet_weights_bytes(int dim, int heads, int layers, int ffn, int precision) {
  Shape sh{1, 64, dim, heads, layers, ffn};
  if (!shape_ok(sh) || layers > 64 || precision != SO3X_PREC_BF16 || !bf16_supported(sh)) return 0;
  return bf16_weights_bytes(sh);
}

int so3x_planenet_prepare(so3x_stream_t s, const float* params, int dim, int heads, int layers, int ffn, int precision, void* weights,
                          size_t weights_bytes) {
  Shape sh{1, 64, dim, heads, layers, ffn};
  if (!shape_ok(sh) || layers > 64 || !params) return SO3X_ERR_INVALID_ARG;
  if (precision != SO3X_PREC_F32 && precision != SO3X_PREC_BF16) return SO3X_ERR_UNSUPPORTED;
  if (precision == SO3X_PREC_F32) return SO3X_OK;                       // the exact form reads the fp32 parameters as they are
  if (!bf16_supported(sh)) return SO3X_ERR_UNSUPPORTED;
  if (!weights || weights_bytes < so3x_planenet_weights_bytes(dim, heads, layers, ffn, precision)) return SO3X_ERR_WORKSPACE;
  return weights_bf16((hipStream_t)s, sh, params, weights, true);
}

int so3x_planenet_fwd(so3x_stream_t s, const float* params, const float* x, const int64_t* t, float* out, float* encoding_out, int64_t B, int64_t P,
                      int dim, int heads, int layers, int ffn, int precision, void* stash, void* workspace, size_t workspace_bytes,
                      const void* prepared_weights, float dropout_p, uint64_t seed, uint64_t rng_offset) {
  Shape sh{B, P, dim, heads, layers, ffn};
  if (!shape_ok(sh) || layers > 64 || (B && (!params || !x || !t || !out))) return SO3X_ERR_INVALID_ARG;
  if (!(dropout_p >= 0.f && dropout_p < 1.f) || (dropout_p > 0.f && !stash)) return SO3X_ERR_INVALID_ARG;   // dropout = a training forward
  if (precision != SO3X_PREC_F32 && precision != SO3X_PREC_BF16) return SO3X_ERR_UNSUPPORTED;
  if (precision == SO3X_PREC_BF16 && !bf16_supported(sh)) return SO3X_ERR_UNSUPPORTED;
  if (B == 0) return SO3X_OK;
  if (!workspace || workspace_bytes < so3x_planenet_workspace_bytes(B, P, dim, heads, layers, ffn, precision)) return SO3X_ERR_WORKSPACE;
  const Drop dr{dropout_p, seed, rng_offset};
  if (precision == SO3X_PREC_BF16) return forward_bf16((hipStream_t)s, sh, params, x, t, out, encoding_out, stash, workspace, prepared_weights, dr);
  const Acts a = stash ? carve_acts(sh, stash, true) : carve_acts(sh, workspace, false);
  // (with a stash the workspace is idle in the forward: the backward's dP buffer holds a layer's dropped-out probabilities)
  return forward_f32((hipStream_t)s, sh, params, x, t, out, encoding_out, a, dr, dr.on() ? carve_bwd(sh, workspace).dprobs : nullptr);
}

int so3x_planenet_bwd(so3x_stream_t s, const float* params, const float* x, const int64_t* t, const float* dout, float* dparams, int64_t B,
                      int64_t P, int dim, int heads, int layers, int ffn, int precision, const void* stash, void* workspace,
                      size_t workspace_bytes, float dropout_p, uint64_t seed, uint64_t rng_offset) {
  Shape sh{B, P, dim, heads, layers, ffn};
  if (!shape_ok(sh) || layers > 64 || !dparams || (B && (!params || !x || !t || !dout || !stash))) return SO3X_ERR_INVALID_ARG;
  if (!(dropout_p >= 0.f && dropout_p < 1.f)) return SO3X_ERR_INVALID_ARG;
  if (precision != SO3X_PREC_F32 && precision != SO3X_PREC_BF16) return SO3X_ERR_UNSUPPORTED;
  if (precision == SO3X_PREC_BF16 && !bf16_supported(sh)) return SO3X_ERR_UNSUPPORTED;
  if (B == 0) {
    hipError_t e = hipMemsetAsync(dparams, 0, (size_t)param_offsets(sh).total * sizeof(float), (hipStream_t)s);
    return e == hipSuccess ? SO3X_OK : (int)e;
  }
  if (!workspace || workspace_bytes < so3x_planenet_workspace_bytes(B, P, dim, heads, layers, ffn, precision)) return SO3X_ERR_WORKSPACE;
  if (precision == SO3X_PREC_BF16) return backward_bf16((hipStream_t)s, sh, params, x, t, dout, dparams, stash, workspace, Drop{dropout_p, seed, rng_offset});
  const Acts a = carve_acts(sh, const_cast<void*>(stash), true);
  const BwdBufs w = carve_bwd(sh, workspace);
  return backward_f32((hipStream_t)s, sh, params, x, dout, dparams, a, w, Drop{dropout_p, seed, rng_offset});
}

}  // extern "C"
