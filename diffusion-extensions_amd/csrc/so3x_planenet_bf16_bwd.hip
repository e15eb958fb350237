// so3x_planenet_bf16_bwd.hip -- backward of the bf16 form of PlaneNet (reference: torch autograd of models.py:185-210, as
// aircraft_rotate.py:104-108 runs it): d sum(out * dout) / d params from the forward's per-layer stash.
//
// Gradients of activations travel in bf16 (as the activations do), parameter gradients are accumulated in fp32.  Kernels:
//   k_gemm_bf16 / k_gemm256_bf16 (so3x_planenet_bf16.hip) for every dX = dY W product, against a TRANSPOSED bf16 image of the
//                    weight (built once per call), with the residual add or the ReLU mask in the epilogue;
//   k_gemm_tn        dW[n][k] = sum over tokens dY[tok][n] X[tok][k]: both operands token-major, so both MFMA operands come
//                    from ds_read_b64_tr_b16; 128 x 128 output tiles, the token axis split over workgroups into fp32 slabs that
//                    k_slab_reduce adds in a fixed order (deterministic, no atomics);
//   k_gemm_tn256     the same product on 256 x 256 tiles (from 32,768 tokens on): fewer transposed reads per MFMA;
//   k_attn_bwd_dq    dQ for 32 queries per wave: S^T, dP^T and dS^T with the query on the lane (as the forward), dQ^T = K^T dS^T
//                    straight from the accumulator registers;
//   k_attn_bwd_dkv   dK, dV for 32 keys per wave: S, dP with the key on the lane, dV^T += dO^T P and dK^T += Q^T dS from the
//                    accumulator registers, Q / dO tiles streamed by LDS-DMA and read by rows AND transposed from one image;
//                    the probabilities are recomputed from the forward's log-sum-exp, never stored;
//   k_ln_bwd_bf16, k_colsum_bf16, k_pool_bwd_tok_bf16, k_embed_bwd_part: rows / reductions (HBM-bound).
#include "so3x_planenet_bf16.hpp"

namespace so3x {
namespace plane {

typedef __attribute__((address_space(3))) s16x4* lds_p;

// ------------------------------------------------------------------------------------------------ transposed weight image
// dst[c][r] = bf16(src[r][c]) for every weight matrix the dX products need, in ONE launch (17 launches of a few microseconds
// each were 4 % of a training evaluation at 32 x 256): 32 x 32 tiles through LDS, the matrix of a block from a table
struct TposeTab {
  int n;
  int tile0[18];      // first block of matrix i (tile0[n] = all blocks)
  int64_t off[17];    // element offset of the matrix in the parameter buffer = in the transposed image
  int R[17], C[17];
};
__global__ __launch_bounds__(256) void k_cvt_bf16_t(const float* __restrict__ prm, bf16* __restrict__ wT, const TposeTab tab) {
  __shared__ float tile[32][33];
  int m = 0;
  while (m + 1 < tab.n && (int)blockIdx.x >= tab.tile0[m + 1]) m++;
  const int R = tab.R[m], C = tab.C[m], tb = blockIdx.x - tab.tile0[m];
  const float* src = prm + tab.off[m];
  bf16* dst = wT + tab.off[m];
  const int c0 = (tb % (C / 32)) * 32, r0 = (tb / (C / 32)) * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
  for (int i = 0; i < 4; i++) tile[ty + 8 * i][tx] = src[(size_t)(r0 + ty + 8 * i) * C + c0 + tx];
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; i++) dst[(size_t)(c0 + ty + 8 * i) * R + r0 + tx] = (bf16)tile[tx][ty + 8 * i];
}

// ------------------------------------------------------------------------------------------------ dW = dY^T X
constexpr int TN256_MIN_TOKENS = 32768;   // below: the 128-wide form (more, smaller workgroups)
// slab[split][n][k] = sum over the split's tokens of Y[tok][n0 + n] X[tok][k0 + k];  Nn % 128 == Kk % 128 == 0, tokens % 64 == 0
// bpart (optional): the column sums of Y (the bias gradient of the same layer) ride along -- a ones vector as a fifth B operand
// gives sum_tok Y[tok][n] in every column of a 16 x 16 accumulator; the k-tile index tk picks which 16-row block of its n range a
// workgroup sums (every workgroup of a row of tiles reads the same Y tile), so the extra MFMA (1 per 16) is spread evenly.
__global__ __launch_bounds__(256, 2) void k_gemm_tn(const bf16* __restrict__ Y, const bf16* __restrict__ X, float* __restrict__ slab, int T,
                                                    int Nn, int Kk, int ldy, int ldx, int tok_per_split, float* __restrict__ bpart) {
  __shared__ __attribute__((aligned(16))) char smem[65536];   // 2 x (Y tile [64 tok][128] 16 KB | X tile 16 KB), swz16 images
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wn = wave >> 1, wk = wave & 1;
  const int nkt = Kk / 128, tn = blockIdx.x / nkt, tk = blockIdx.x % nkt, split = blockIdx.y;
  const int t0 = split * tok_per_split, t1 = t0 + tok_per_split < T ? t0 + tok_per_split : T;
  const int nt = t1 > t0 ? (t1 - t0) / 64 : 0;
  const bf16* Yg = Y + (size_t)t0 * ldy + tn * 128;
  const bf16* Xg = X + (size_t)t0 * ldx + tk * 128;
  auto stage = [&](int j, int buf) {
    char* sy = smem + buf * 32768;
    char* sx = sy + 16384;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int rowblk = (wave * 4 + i) * 4, row = rowblk + (lane >> 4);
      const int ch = (lane & 15) ^ swz16(row);
      glds16_asm(Yg + (size_t)(j * 64 + row) * ldy + ch * 8, sy + rowblk * 256);
      glds16_asm(Xg + (size_t)(j * 64 + row) * ldx + ch * 8, sx + rowblk * 256);
    }
  };
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 accb = {0.f, 0.f, 0.f, 0.f};
  const int nke = nkt < 4 ? nkt : 4;
  const bool sums = bpart != nullptr && wk == 0 && tk < nke;   // this wave sums the blocks i with i % nke == tk
  const bf16x8 ones = {(bf16)1.f, (bf16)1.f, (bf16)1.f, (bf16)1.f, (bf16)1.f, (bf16)1.f, (bf16)1.f, (bf16)1.f};
  // transposed-read lane constants: the 16-lane group g reads tokens 8 g + (0..3) [+4] of a k-step, lane 4 q + p of the group
  // supplies row q, columns 4 p .. 4 p + 3 of the 16-column block
  const int g = lane >> 4, q_ = (lane & 15) >> 2, p_ = lane & 3;
  const int sub = (p_ & 1) * 8, clo = p_ >> 1;
  const int row0 = 8 * g + q_;                          // + 32 s (+ 4)
  const int sw0 = swz16(row0), sw1 = swz16(row0 + 4);   // swz16 does not see + 32 s
  if (nt > 0) stage(0, 0);
  for (int j = 0; j < nt; j++) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (j + 1 < nt) stage(j + 1, (j + 1) & 1);
    const char* sy = smem + (j & 1) * 32768;
    const char* sx = sy + 16384;
#pragma unroll
    for (int s = 0; s < 2; s++) {
      bf16x8 a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const int cy = 2 * (wn * 4 + i) + clo, cx = 2 * (wk * 4 + i) + clo;   // 16-byte chunk of the block's columns
        const char* ry = sy + (32 * s + row0) * 256;
        const char* rx = sx + (32 * s + row0) * 256;
        const s16x4 y0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(ry + ((cy ^ sw0) << 4) + sub));
        const s16x4 y1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(ry + 1024 + ((cy ^ sw1) << 4) + sub));
        const s16x4 x0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(rx + ((cx ^ sw0) << 4) + sub));
        const s16x4 x1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(rx + 1024 + ((cx ^ sw1) << 4) + sub));
        a[i] = __builtin_bit_cast(bf16x8, (s16x8){y0[0], y0[1], y0[2], y0[3], y1[0], y1[1], y1[2], y1[3]});
        b[i] = __builtin_bit_cast(bf16x8, (s16x8){x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]});
      }
#pragma unroll
      for (int i = 0; i < 4; i++)
#pragma unroll
        for (int jj = 0; jj < 4; jj++) acc[i][jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[jj], acc[i][jj], 0, 0, 0);
      if (sums) {
        if (nke == 4) {   // one block per workgroup: a[tk]
          const bf16x8 at = tk == 0 ? a[0] : (tk == 1 ? a[1] : (tk == 2 ? a[2] : a[3]));
          accb = __builtin_amdgcn_mfma_f32_16x16x32_bf16(at, ones, accb, 0, 0, 0);
        }
      }
    }
  }
  if (sums && nke == 4 && (lane & 15) == 0) {
    float* bo = bpart + (size_t)split * Nn + tn * 128 + wn * 64 + tk * 16 + g * 4;
#pragma unroll
    for (int e = 0; e < 4; e++) bo[e] = accb[e];
  }
  float* out = slab + (size_t)split * Nn * Kk;
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int jj = 0; jj < 4; jj++)
#pragma unroll
      for (int e = 0; e < 4; e++)
        out[(size_t)(tn * 128 + wn * 64 + i * 16 + g * 4 + e) * Kk + tk * 128 + wk * 64 + jj * 16 + (lane & 15)] = acc[i][jj][e];
}
__global__ __launch_bounds__(256) void k_slab_reduce(const float* __restrict__ slab, int nsplit, int64_t n4, float* __restrict__ out,
                                                     const float* __restrict__ bpart, int nb4, float* __restrict__ bout) {
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) {   // the bias partials behind the matrix: the same fixed order over the splits
    i -= n4;
    if (i >= nb4) return;
    float4 a = reinterpret_cast<const float4*>(bpart)[i];
    for (int s = 1; s < nsplit; s++) {
      const float4 v = reinterpret_cast<const float4*>(bpart + (size_t)s * nb4 * 4)[i];
      a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
    reinterpret_cast<float4*>(bout)[i] = a;   // (parameter offsets are multiples of 4 floats: every matrix has a multiple of 4 rows)
    return;
  }
  float4 a = reinterpret_cast<const float4*>(slab)[i];
  for (int s = 1; s < nsplit; s++) {
    const float4 v = reinterpret_cast<const float4*>(slab + (size_t)s * n4 * 4)[i];
    a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
  }
  reinterpret_cast<float4*>(out)[i] = a;
}
// ---- the large-T form: 256 x 256 output tiles, 8 waves (2 along n x 4 along k, 128 x 64 of the tile each) -------------------------
// The 128-wide kernel is bound by its transposed LDS reads: two ds_read_b64_tr_b16 (4 clocks each) per fragment and one fragment per
// MFMA and operand -- 64 LDS clocks per 32-token step and wave, 512 for the 8 waves of a CU, against 512 matrix-pipe clocks per SIMD:
// the LDS would have to be busy every clock for the matrix pipe to be (measured: 0.70 PFLOP/s, pipe 28-36 % busy).  A wave that
// owns 128 x 64 of a 256 x 256 tile reads 12 fragments for 32 MFMAs (0.75 reads per MFMA instead of 2).  K tiles are 32 tokens
// (Y [32][256] | X [32][256] = 32 KB) through a FOUR-stage ring: three stages in flight behind the one being read (a 32-token
// step is 0.43 us of MFMAs per SIMD; one stage ahead would expose the load round trip at one workgroup per CU), counted waits.
// LDS rows are 512 bytes, chunk c of a row at c ^ swz16(row): the swizzle only moves a chunk inside its 256-byte half, where it
// spreads the four rows a 16-lane group reads over the banks exactly as in the 128-wide image.
// NKE (0 = no bias): the bias gradient's column sums ride along as in k_gemm_tn -- the n-block i of a wave (16 rows) is summed by the
// workgroup of k-tile i % NKE, NKE = min(Kk / 256, 8).
template <int NKE>
__global__ __launch_bounds__(512, 1) void k_gemm_tn256(const bf16* __restrict__ Y, const bf16* __restrict__ X, float* __restrict__ slab, int T,
                                                       int Nn, int Kk, int ldy, int ldx, int tok_per_split, float* __restrict__ bpart) {
  extern __shared__ __attribute__((aligned(16))) char smem[];   // 4 x (Y tile [32 tok][256] 16 KB | X tile 16 KB)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wn = wave >> 2, wk = wave & 3;
  const int nkt = Kk / 256, tn = blockIdx.x / nkt, tk = blockIdx.x % nkt, split = blockIdx.y;
  const int t0 = split * tok_per_split, t1 = t0 + tok_per_split < T ? t0 + tok_per_split : T;
  const int nt = t1 > t0 ? (t1 - t0) / 32 : 0;
  const bf16* Yg = Y + (size_t)t0 * ldy + tn * 256;
  const bf16* Xg = X + (size_t)t0 * ldx + tk * 256;
  // staging: one LDS-DMA instruction = two 512-byte rows; a stage's 16 row pairs per operand go two to a wave
  const int srow = lane >> 5, spos = lane & 31;
  auto stage = [&](int j, int buf) {
    char* sy = smem + buf * 32768;
    char* sx = sy + 16384;
#pragma unroll
    for (int i = 0; i < 2; i++) {
      const int rowpair = wave * 2 + i, row = 2 * rowpair + srow;
      const int ch = spos ^ swz16(row);
      glds16_asm(Yg + (size_t)(j * 32 + row) * ldy + ch * 8, sy + rowpair * 1024);
      glds16_asm(Xg + (size_t)(j * 32 + row) * ldx + ch * 8, sx + rowpair * 1024);
    }
  };
  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  constexpr int NB = NKE ? 8 / NKE : 1;
  f32x4 accb[NB];
#pragma unroll
  for (int i = 0; i < NB; i++) accb[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  const bool sums = NKE != 0 && bpart != nullptr && wk == 0 && tk < NKE;
  const bf16x8 ones = {(bf16)1.f, (bf16)1.f, (bf16)1.f, (bf16)1.f, (bf16)1.f, (bf16)1.f, (bf16)1.f, (bf16)1.f};
  // transposed-read lane constants (as k_gemm_tn): the 16-lane group g reads tokens 8 g + (0..3) [+4], lane 4 q + p of the group
  // supplies row q, columns 4 p .. 4 p + 3 of the 16-column block
  const int g = lane >> 4, q_ = (lane & 15) >> 2, p_ = lane & 3;
  const int sub = (p_ & 1) * 8, clo = p_ >> 1;
  const int row0 = 8 * g + q_;
  const int sw0 = swz16(row0), sw1 = swz16(row0 + 4);
  for (int u = 0; u < 3; u++)
    if (u < nt) stage(u, u);
  for (int j = 0; j < nt; j++) {
    // stage j has landed; up to two younger stages (four DMAs per wave each) stay in flight
    const int younger = nt - 1 - j;
    if (younger >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (younger == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                   // ... everyone's has, and everyone is done reading stage j - 1's buffer
    if (j + 3 < nt) stage(j + 3, (j + 3) & 3);
    const char* ry = smem + (j & 3) * 32768 + row0 * 512;
    const char* rx = ry + 16384;
    bf16x8 a[8], b[4];
#pragma unroll
    for (int i = 0; i < 8; i++) {
      const int cy = 2 * (wn * 8 + i) + clo;
      const s16x4 y0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(ry + ((cy ^ sw0) << 4) + sub));
      const s16x4 y1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(ry + 2048 + ((cy ^ sw1) << 4) + sub));
      a[i] = __builtin_bit_cast(bf16x8, (s16x8){y0[0], y0[1], y0[2], y0[3], y1[0], y1[1], y1[2], y1[3]});
    }
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int cx = 2 * (wk * 4 + i) + clo;
      const s16x4 x0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(rx + ((cx ^ sw0) << 4) + sub));
      const s16x4 x1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(rx + 2048 + ((cx ^ sw1) << 4) + sub));
      b[i] = __builtin_bit_cast(bf16x8, (s16x8){x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]});
    }
#pragma unroll
    for (int i = 0; i < 8; i++)
#pragma unroll
      for (int jj = 0; jj < 4; jj++) acc[i][jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[jj], acc[i][jj], 0, 0, 0);
    if constexpr (NKE != 0) {
      if (sums) {
#pragma unroll
        for (int i = 0; i < 8; i++)
          if (i % NKE == tk) accb[i / NKE] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], ones, accb[i / NKE], 0, 0, 0);
      }
    }
  }
  if constexpr (NKE != 0) {
    if (sums && (lane & 15) == 0) {
#pragma unroll
      for (int i = 0; i < 8; i++)
        if (i % NKE == tk) {
          float* bo = bpart + (size_t)split * Nn + tn * 256 + wn * 128 + i * 16 + g * 4;
#pragma unroll
          for (int e = 0; e < 4; e++) bo[e] = accb[i / NKE][e];
        }
    }
  }
  // The tile goes out through LDS, 128 of its rows at a time (all 128 KB: [row][256] fp32, the row's 16-column block c at c ^
  // ((row >> 2) & 15) so that the four row groups of an accumulator store land 64 bytes apart), and leaves as whole 1 KB rows of
  // float4 stores.  (Straight from the accumulators every store instruction wrote four 64-byte pieces of four different rows:
  // 64 MB of slabs per product at a fraction of the write rate.)
  float* out = slab + (size_t)split * Nn * Kk + (size_t)(tn * 256) * Kk + tk * 256;
  float* sc = reinterpret_cast<float*>(smem);
#pragma unroll
  for (int hh = 0; hh < 2; hh++) {
    __syncthreads();   // the ring (first pass) / the previous pass's rows have been read
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
      for (int jj = 0; jj < 4; jj++)
#pragma unroll
        for (int e = 0; e < 4; e++) {
          const int row = wn * 64 + i * 16 + g * 4 + e, cb = wk * 4 + jj;     // row of this pass's 128, 16-column block of 16
          sc[row * 256 + ((cb ^ ((row >> 2) & 15)) << 4) + (lane & 15)] = acc[4 * hh + i][jj][e];
        }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 16; it++) {
      const int row = it * 8 + (tid >> 6), c4 = tid & 63;                     // float4 c4 of the row: block c4 >> 2, quarter c4 & 3
      const float4 v = *reinterpret_cast<const float4*>(sc + row * 256 + (((c4 >> 2) ^ ((row >> 2) & 15)) << 4) + (c4 & 3) * 4);
      const int trow = (row >> 6) * 128 + hh * 64 + (row & 63);               // tile row: wave row group wn, pass hh
      *reinterpret_cast<float4*>(out + (size_t)trow * Kk + c4 * 4) = v;
    }
  }
}
constexpr int TN256_LDS = 4 * 32768;
static PerDevice g_tn256[4];

constexpr int64_t SLAB_FLOATS = (int64_t)16 * FF * D;  // room for 16 slabs of the largest matrix (or 64 of a 512 x 512 one)
constexpr int64_t BIAS_PART_FLOATS = (int64_t)64 * FF; // behind them: up to 64 splits of the widest bias
// dW[Nn][Kk] (fp32, overwritten) = Y[:T][:Nn]^T X[:T][:Kk].  The token axis is cut into as many slabs as keep ~512 workgroups busy
// (2 per CU; the 256-wide form: 256, one per CU) and fit the slab buffer; the cut depends on the shapes only, so the summation order
// is fixed.  dbias (optional, [Nn]) = the column sums of Y, from the same pass (Kk >= 512: the k-tiles to spread them over).
static int gemm_tn(hipStream_t s, const bf16* Y, int ldy, const bf16* X, int ldx, float* dW, int T, int Nn, int Kk, float* slab,
                   float* dbias = nullptr) {
  if (dbias && Kk < 512) return SO3X_ERR_INVALID_ARG;
  float* bpart = dbias ? slab + SLAB_FLOATS : nullptr;
  if (Nn % 128 || Kk % 128 || T % 64) return SO3X_ERR_INVALID_ARG;
  const int64_t n4 = (int64_t)Nn * Kk / 4;
  const int nb4 = dbias ? Nn / 4 : 0;
  int nsplit;
  const int nkt256 = Kk / 256;
  if (Nn % 256 == 0 && Kk % 256 == 0 && T >= TN256_MIN_TOKENS && (!dbias || nkt256 == 2 || nkt256 == 4 || nkt256 >= 8)) {
    const int tiles = (Nn / 256) * nkt256;
    nsplit = (256 + tiles - 1) / tiles;
    if (nsplit > 64) nsplit = 64;
    if ((int64_t)nsplit * Nn * Kk > SLAB_FLOATS) nsplit = (int)(SLAB_FLOATS / ((int64_t)Nn * Kk));
    if (nsplit > T / 128) nsplit = T / 128;
    if (nsplit < 1) nsplit = 1;
    const int per = ((T / 32 + nsplit - 1) / nsplit) * 32;
    const int nke = !dbias ? 0 : (nkt256 >= 8 ? 8 : nkt256);
#define SO3X_TN256(NKE, SLOT)                                                                                                         \
  {                                                                                                                                   \
    if (int rc = ensure_dyn_lds(g_tn256[SLOT], (const void*)k_gemm_tn256<NKE>, TN256_LDS)) return rc;                                  \
    hipLaunchKernelGGL(k_gemm_tn256<NKE>, dim3(tiles, nsplit), dim3(512), TN256_LDS, s, Y, X, slab, T, Nn, Kk, ldy, ldx, per, bpart);  \
  }
    if (nke == 0) SO3X_TN256(0, 0)
    else if (nke == 2) SO3X_TN256(2, 1)
    else if (nke == 4) SO3X_TN256(4, 2)
    else SO3X_TN256(8, 3)
#undef SO3X_TN256
  } else {
    const int tiles = (Nn / 128) * (Kk / 128);
    nsplit = (512 + tiles - 1) / tiles;
    if (nsplit > 32) nsplit = 32;
    if ((int64_t)nsplit * Nn * Kk > SLAB_FLOATS) nsplit = (int)(SLAB_FLOATS / ((int64_t)Nn * Kk));
    if (nsplit > T / 64) nsplit = T / 64;
    if (nsplit < 1) nsplit = 1;
    const int per = ((T / 64 + nsplit - 1) / nsplit) * 64;
    hipLaunchKernelGGL(k_gemm_tn, dim3(tiles, nsplit), dim3(256), 0, s, Y, X, slab, T, Nn, Kk, ldy, ldx, per, bpart);
  }
  hipLaunchKernelGGL(k_slab_reduce, dim3((unsigned)((n4 + nb4 + 255) / 256)), dim3(256), 0, s, slab, nsplit, n4, dW, bpart, nb4, dbias);
  return check_launch();
}

// ------------------------------------------------------------------------------------------------ attention backward
// delta[b][h][q] = sum_d dO[q][d] O[q][d]   (the softmax backward's row constant)
//   lse2 = lse * log2(e): what the recomputed exponentials subtract
__global__ __launch_bounds__(256) void k_attn_delta(const bf16* __restrict__ o, const bf16* __restrict__ dO, const float* __restrict__ lse,
                                                    float* __restrict__ delta, float* __restrict__ lse2, int64_t N, int P) {
  const int64_t idx = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);   // (token, head): 16 lanes x 8 columns = one head's 128
  if (idx >= N * HEADS) return;
  const int64_t n = idx / HEADS;
  const int hd = (int)(idx - n * HEADS), l16 = threadIdx.x & 15;
  const bf16x8 a = *reinterpret_cast<const bf16x8*>(o + n * D + hd * DH + l16 * 8);
  const bf16x8 b = *reinterpret_cast<const bf16x8*>(dO + n * D + hd * DH + l16 * 8);
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; i++) s = fmaf((float)a[i], (float)b[i], s);
#pragma unroll
  for (int off = 8; off > 0; off >>= 1) s += __shfl_xor(s, off);
  if (l16 == 0) {
    const int64_t b_ = n / P, q = n - b_ * P;
    delta[(b_ * HEADS + hd) * P + q] = s;
    lse2[(b_ * HEADS + hd) * P + q] = lse[(b_ * HEADS + hd) * P + q] * 1.4426950408889634f;
  }
}

// dQ: one wave = 32 queries (lane pair per query, as the forward), 64-key tiles of K and V by LDS-DMA.
//   S^T = K Q^T, P^T = exp(S^T / sqrt(dh) - lse_q), dP^T = V dO^T, dS^T = P^T (dP^T - delta_q) / sqrt(dh), dQ^T += K^T dS^T
// DROP (all three kernels): the forward ran with dropout on the probabilities; dP is masked and rescaled (dQ, dK), P is masked and
// dV rescaled -- keep bits from the forward's maskq (query on the lane) / maskk (key on the lane), staged per tile like the operands
template <bool DROP>
__global__ __launch_bounds__(256, 2) void k_attn_bwd_dq(const bf16* __restrict__ qkv, const bf16* __restrict__ dO, const float* __restrict__ lse2v,
                                                        const float* __restrict__ delta, bf16* __restrict__ dqkv, int P, float sc, float c2,
                                                        const uint32_t* __restrict__ maskq, float inv_keep) {
  __shared__ __attribute__((aligned(16))) char smem[65536 + (DROP ? 2048 : 0)];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), r = lane & 31, h = lane >> 5;
  int xt, hd, b;
  attn_tile((P + 127) / 128, xt, hd, b);
  const int q0 = xt * 128 + wave * 32;
  const size_t tok0 = (size_t)b * P;
  const int qr = q0 + r < P ? q0 + r : P - 1;
  bf16x8 qf[8], dof[8];
  {
    const bf16* qrow = qkv + (tok0 + qr) * (3 * D) + hd * DH + 8 * h;
    const bf16* drow = dO + (tok0 + qr) * D + hd * DH + 8 * h;
#pragma unroll
    for (int ks = 0; ks < 8; ks++) {
      qf[ks] = *reinterpret_cast<const bf16x8*>(qrow + 16 * ks);
      dof[ks] = *reinterpret_cast<const bf16x8*>(drow + 16 * ks);
    }
  }
  const float lse2 = lse2v[((size_t)b * HEADS + hd) * P + qr];
  const float dl = delta[((size_t)b * HEADS + hd) * P + qr];
  const bf16* kbase = qkv + tok0 * (3 * D) + D + hd * DH;
  const bf16* vbase = kbase + D;
  const uint32_t* mrow = nullptr;
  if constexpr (DROP) mrow = maskq + (((size_t)b * HEADS + hd) * P + qr) * (P / 32) + h;
  auto stage = [&](int j, int buf) {
    char* sk = smem + buf * 32768;
    char* sv = sk + 16384;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int rowblk = (wave * 4 + i) * 4, row = rowblk + (lane >> 4);
      const int ch = (lane & 15) ^ swz16(row);
      const size_t off = (size_t)(j * 64 + row) * (3 * D) + ch * 8;
      glds16_asm(kbase + off, sk + rowblk * 256);
      glds16_asm(vbase + off, sv + rowblk * 256);
    }
    if constexpr (DROP) glds4_asm(mrow + 2 * j, smem + 65536 + buf * 1024 + wave * 256);
  };
  f32x16 dq[4];
#pragma unroll
  for (int dt = 0; dt < 4; dt++)
#pragma unroll
    for (int i = 0; i < 16; i++) dq[dt][i] = 0.f;
  const int krd = r * 256, ksw = swz16(r);
  const int g = lane >> 4, q_ = (lane & 15) >> 2, p_ = lane & 3;
  const int vrow0 = (4 * h + q_) * 256, vsub = (p_ & 1) * 8, vclo = 2 * (g & 1) + (p_ >> 1);
  const int nt = P / 64;
  stage(0, 0);
  for (int j = 0; j < nt; j++) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (j + 1 < nt) stage(j + 1, (j + 1) & 1);
    const char* sk = smem + (j & 1) * 32768;
    const char* sv = sk + 16384;
    f32x16 st[2], dp[2];
#pragma unroll
    for (int kb = 0; kb < 2; kb++) {
#pragma unroll
      for (int i = 0; i < 16; i++) { st[kb][i] = 0.f; dp[kb][i] = 0.f; }
#pragma unroll
      for (int ks = 0; ks < 8; ks++) {
        const int off = kb * 8192 + krd + (((2 * ks + h) ^ ksw) << 4);
        st[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(sk + off), qf[ks], st[kb], 0, 0, 0);
        dp[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(sv + off), dof[ks], dp[kb], 0, 0, 0);
      }
    }
#pragma unroll
    for (int kb = 0; kb < 2; kb++) {
      uint32_t wsh = 0u;
      if constexpr (DROP) wsh = reinterpret_cast<const uint32_t*>(smem + 65536 + (j & 1) * 1024 + wave * 256)[r + 32 * kb] >> (4 * h);
#pragma unroll
      for (int i = 0; i < 16; i++) {
        const float pv = __builtin_amdgcn_exp2f(fmaf(st[kb][i], c2, -lse2));
        float dpv = dp[kb][i];
        if constexpr (DROP) dpv = (wsh >> ((i & 3) + 8 * (i >> 2))) & 1u ? dpv * inv_keep : 0.f;
        st[kb][i] = pv * (dpv - dl) * sc;
      }
    }
#pragma unroll
    for (int s4 = 0; s4 < 4; s4++) {
      const int kb = s4 >> 1, s1 = s4 & 1;
      const bf16x8 dsf = {(bf16)st[kb][8 * s1 + 0], (bf16)st[kb][8 * s1 + 1], (bf16)st[kb][8 * s1 + 2], (bf16)st[kb][8 * s1 + 3],
                          (bf16)st[kb][8 * s1 + 4], (bf16)st[kb][8 * s1 + 5], (bf16)st[kb][8 * s1 + 6], (bf16)st[kb][8 * s1 + 7]};
#pragma unroll
      for (int dt = 0; dt < 4; dt++) {
        const int c0 = ((((dt ^ q_) << 2) | (vclo ^ h)) << 4) + vsub;
        const int c1 = ((((dt ^ q_) << 2) | (vclo ^ (h + 2))) << 4) + vsub;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(sk + s4 * 4096 + vrow0 + c0));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(sk + s4 * 4096 + vrow0 + 2048 + c1));
        const s16x8 v8 = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        dq[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, v8), dsf, dq[dt], 0, 0, 0);
      }
    }
  }
  if (q0 >= P) return;
  bf16* orow = dqkv + (tok0 + q0 + r) * (3 * D) + hd * DH + 4 * h;
#pragma unroll
  for (int dt = 0; dt < 4; dt++)
#pragma unroll
    for (int g4 = 0; g4 < 4; g4++)
      *reinterpret_cast<bf16x4*>(orow + 32 * dt + 8 * g4) =
          bf16x4{(bf16)dq[dt][4 * g4], (bf16)dq[dt][4 * g4 + 1], (bf16)dq[dt][4 * g4 + 2], (bf16)dq[dt][4 * g4 + 3]};
}

// dK, dV: one wave = 32 keys (key on the lane; K and V rows of the wave in registers as B operands), 64-query tiles of Q and dO by
// LDS-DMA.  Per 32-query block:
//   S = Q K^T, P = exp(S / sqrt(dh) - lse_q), dP = dO V^T, dS = P (dP - delta_q) / sqrt(dh)     (query = accumulator row)
//   dV^T += dO^T P, dK^T += Q^T dS    (A operands = transposed reads of the same dO / Q tiles, B = the accumulators as they sit)
// (One launch per output: holding dK^T AND dV^T -- 128 accumulator registers -- beside both operand sets does not fit 256 registers;
//  WHICH = 0: dV, from S and P only; WHICH = 1: dK, from S, dP and dS.)
template <int WHICH, bool DROP>
__global__ __launch_bounds__(256, 2) void k_attn_bwd_dkv(const bf16* __restrict__ qkv, const bf16* __restrict__ dO, const float* __restrict__ lse2v,
                                                         const float* __restrict__ delta, bf16* __restrict__ dqkv, int P, float sc, float c2,
                                                         const uint32_t* __restrict__ maskk, float inv_keep) {
  // 2 x (Q tile 16 KB | dO tile 16 KB) | 2 x (lse2[64] | delta[64]) [| 2 x 4 waves x 64 mask words]
  __shared__ __attribute__((aligned(16))) char smem[65536 + 1024 + (DROP ? 2048 : 0)];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), r = lane & 31, h = lane >> 5;
  int xt, hd, b;
  attn_tile((P + 127) / 128, xt, hd, b);
  const int k0 = xt * 128 + wave * 32;
  const size_t tok0 = (size_t)b * P;
  const int kr = k0 + r < P ? k0 + r : P - 1;
  bf16x8 kf[8], vf[WHICH ? 8 : 1];
  {
    const bf16* krow = qkv + (tok0 + kr) * (3 * D) + D + hd * DH + 8 * h;
#pragma unroll
    for (int ks = 0; ks < 8; ks++) {
      kf[ks] = *reinterpret_cast<const bf16x8*>(krow + 16 * ks);
      if constexpr (WHICH == 1) vf[ks] = *reinterpret_cast<const bf16x8*>(krow + D + 16 * ks);
    }
  }
  const bf16* qbase = qkv + tok0 * (3 * D) + hd * DH;
  const bf16* dbase = dO + tok0 * D + hd * DH;
  const float* lrow = lse2v + ((size_t)b * HEADS + hd) * P;
  const float* drow = delta + ((size_t)b * HEADS + hd) * P;
  const uint32_t* mrow = nullptr;   // lane (r, h): word 2 j + h of key r's row of keep bits (one bit per query)
  if constexpr (DROP) mrow = maskk + (((size_t)b * HEADS + hd) * P + kr) * (P / 32) + h;
  auto stage = [&](int j, int buf) {
    char* sq = smem + buf * 32768;
    char* sd = sq + 16384;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int rowblk = (wave * 4 + i) * 4, row = rowblk + (lane >> 4);
      const int ch = (lane & 15) ^ swz16(row);
      glds16_asm(qbase + (size_t)(j * 64 + row) * (3 * D) + ch * 8, sq + rowblk * 256);
      glds16_asm(dbase + (size_t)(j * 64 + row) * D + ch * 8, sd + rowblk * 256);
    }
    // the tile's 64 lse2 and 64 delta values by LDS-DMA too (4 bytes per lane: no register-destination load beside the DMA)
    char* sl = smem + 65536 + buf * 512;
    if (wave == 0) glds4_asm(lrow + j * 64 + lane, sl);
    if (wave == 1) glds4_asm(drow + j * 64 + lane, sl + 256);
    if constexpr (DROP) glds4_asm(mrow + 2 * j, smem + 65536 + 1024 + buf * 1024 + wave * 256);
  };
  f32x16 acc[4];   // dV^T or dK^T: [d][key]
#pragma unroll
  for (int dt = 0; dt < 4; dt++)
#pragma unroll
    for (int i = 0; i < 16; i++) acc[dt][i] = 0.f;
  const int qrd = r * 256, qsw = swz16(r);
  const int g = lane >> 4, q_ = (lane & 15) >> 2, p_ = lane & 3;
  const int trow0 = (4 * h + q_) * 256, tsub = (p_ & 1) * 8, tclo = 2 * (g & 1) + (p_ >> 1);
  const int nt = P / 64;
  stage(0, 0);
  for (int j = 0; j < nt; j++) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (j + 1 < nt) stage(j + 1, (j + 1) & 1);
    const char* sq = smem + (j & 1) * 32768;
    const char* sd = sq + 16384;
    const float* sl = reinterpret_cast<const float*>(smem + 65536 + (j & 1) * 512);
#pragma unroll
    for (int qb = 0; qb < 2; qb++) {
      f32x16 s_, dp;
#pragma unroll
      for (int i = 0; i < 16; i++) { s_[i] = 0.f; dp[i] = 0.f; }
#pragma unroll
      for (int ks = 0; ks < 8; ks++) {
        const int off = qb * 8192 + qrd + (((2 * ks + h) ^ qsw) << 4);
        s_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(sq + off), kf[ks], s_, 0, 0, 0);
        if constexpr (WHICH == 1) dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(sd + off), vf[ks], dp, 0, 0, 0);
      }
      // accumulator register i = query row (i & 3) + 8 (i >> 2) + 4 h of the block: its lse / delta from the tile's LDS copy
      uint32_t wsh = 0u;
      if constexpr (DROP) wsh = reinterpret_cast<const uint32_t*>(smem + 65536 + 1024 + (j & 1) * 1024 + wave * 256)[r + 32 * qb] >> (4 * h);
#pragma unroll
      for (int g4 = 0; g4 < 4; g4++) {
        const float4 l4 = *reinterpret_cast<const float4*>(sl + qb * 32 + 8 * g4 + 4 * h);
        const float4 d4 = *reinterpret_cast<const float4*>(sl + 64 + qb * 32 + 8 * g4 + 4 * h);
        const float lv[4] = {l4.x, l4.y, l4.z, l4.w}, dv[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
        for (int e = 0; e < 4; e++) {
          const int i = 4 * g4 + e;
          const float pv = __builtin_amdgcn_exp2f(fmaf(s_[i], c2, -lv[e]));
          bool keep = true;
          if constexpr (DROP) keep = (wsh >> ((i & 3) + 8 * (i >> 2))) & 1u;
          if constexpr (WHICH == 1) s_[i] = pv * ((DROP ? (keep ? dp[i] * inv_keep : 0.f) : dp[i]) - dv[e]) * sc;   // dS
          else s_[i] = keep ? pv : 0.f;                                                                            // (kept) P
        }
      }
#pragma unroll
      for (int s1 = 0; s1 < 2; s1++) {
        const bf16x8 pf = {(bf16)s_[8 * s1 + 0], (bf16)s_[8 * s1 + 1], (bf16)s_[8 * s1 + 2], (bf16)s_[8 * s1 + 3],
                           (bf16)s_[8 * s1 + 4], (bf16)s_[8 * s1 + 5], (bf16)s_[8 * s1 + 6], (bf16)s_[8 * s1 + 7]};
        const int s4 = qb * 2 + s1;      // queries 16 s4 .. 16 s4 + 15 of the tile
#pragma unroll
        for (int dt = 0; dt < 4; dt++) {
          const int c0 = ((((dt ^ q_) << 2) | (tclo ^ h)) << 4) + tsub;
          const int c1 = ((((dt ^ q_) << 2) | (tclo ^ (h + 2))) << 4) + tsub;
          const char* st = WHICH == 1 ? sq : sd;      // dK^T += Q^T dS; dV^T += dO^T P
          const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(st + s4 * 4096 + trow0 + c0));
          const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(st + s4 * 4096 + trow0 + 2048 + c1));
          acc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
              __builtin_bit_cast(bf16x8, (s16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]}), pf, acc[dt], 0, 0, 0);
        }
      }
    }
  }
  if (k0 >= P) return;
  bf16* krow = dqkv + (tok0 + k0 + r) * (3 * D) + (WHICH == 1 ? D : 2 * D) + hd * DH + 4 * h;
  const float os = (DROP && WHICH == 0) ? inv_keep : 1.f;   // dV = (kept P / keep)^T dO
#pragma unroll
  for (int dt = 0; dt < 4; dt++)
#pragma unroll
    for (int g4 = 0; g4 < 4; g4++)
      *reinterpret_cast<bf16x4*>(krow + 32 * dt + 8 * g4) =
          bf16x4{(bf16)(acc[dt][4 * g4] * os), (bf16)(acc[dt][4 * g4 + 1] * os), (bf16)(acc[dt][4 * g4 + 2] * os), (bf16)(acc[dt][4 * g4 + 3] * os)};
}

// ------------------------------------------------------------------------------------------------ rows and reductions
// LayerNorm backward over 512-wide bf16 rows: dr = rstd (dy gamma - mean(dy gamma) - xhat mean(dy gamma xhat)); per workgroup (64
// rows) the partial column sums part[blk][0..511] = sum dy xhat (d gamma), part[blk][512..1023] = sum dy (d beta)
// rows per workgroup: 64 at large token counts (fewer partial rows to add up), 16 below 32768 tokens (a wave's rows are a serial
// chain of load -> two wave reductions -> store: 128 workgroups x 16 rows per wave took 18 us for 8 MB at 32 x 256)
inline int lnb_rows(int64_t Np) { return Np >= 32768 ? 64 : 16; }
// drm (optional): dr through the dropout of the block whose output the LayerNorm's input added to its residual (the gradient that
// block sees; dr itself goes on along the residual)
__global__ __launch_bounds__(256) void k_ln_bwd_bf16(const bf16* __restrict__ dy, const bf16* __restrict__ r, const float* __restrict__ stats,
                                                     const float* __restrict__ gamma, bf16* __restrict__ dr, float* __restrict__ part, int64_t rows,
                                                     bf16* __restrict__ drm, const GemmDrop gd, int LNB_ROWS) {
  __shared__ float red[4][1024];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float4 g0 = *reinterpret_cast<const float4*>(gamma + lane * 8), g1 = *reinterpret_cast<const float4*>(gamma + lane * 8 + 4);
  const float gm[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
  float dg[8], db[8];
#pragma unroll
  for (int i = 0; i < 8; i++) { dg[i] = 0.f; db[i] = 0.f; }
  const int64_t r0 = (int64_t)blockIdx.x * LNB_ROWS;
  for (int k = wave; k < LNB_ROWS; k += 4) {
    const int64_t row = r0 + k;
    if (row >= rows) break;
    const bf16x8 dv = *reinterpret_cast<const bf16x8*>(dy + row * D + lane * 8);
    const bf16x8 rv = *reinterpret_cast<const bf16x8*>(r + row * D + lane * 8);
    const float mean = stats[row * 2], rstd = stats[row * 2 + 1];
    float gy[8], xh[8], c1 = 0.f, c2 = 0.f;
#pragma unroll
    for (int i = 0; i < 8; i++) {
      const float d = (float)dv[i];
      xh[i] = ((float)rv[i] - mean) * rstd;
      gy[i] = d * gm[i];
      c1 += gy[i];
      c2 = fmaf(gy[i], xh[i], c2);
      dg[i] = fmaf(d, xh[i], dg[i]);
      db[i] += d;
    }
    c1 = wave_sum(c1) * (1.f / D);
    c2 = wave_sum(c2) * (1.f / D);
    bf16x8 out;
    float ov[8];
#pragma unroll
    for (int i = 0; i < 8; i++) { ov[i] = rstd * (gy[i] - c1 - xh[i] * c2); out[i] = (bf16)ov[i]; }
    *reinterpret_cast<bf16x8*>(dr + row * D + lane * 8) = out;
    if (drm) {
      const uint32_t m = drop_keep8(gd.seed, (uint64_t)row * (D / 8) + lane, gd.ctr_hi, gd.thr16);
#pragma unroll
      for (int i = 0; i < 8; i++) out[i] = (m >> i) & 1u ? (bf16)(ov[i] * gd.scale) : (bf16)0.f;
      *reinterpret_cast<bf16x8*>(drm + row * D + lane * 8) = out;
    }
  }
#pragma unroll
  for (int i = 0; i < 8; i++) {
    red[wave][lane * 8 + i] = dg[i];
    red[wave][512 + lane * 8 + i] = db[i];
  }
  __syncthreads();
  for (int c = threadIdx.x; c < 1024; c += 256) part[(int64_t)blockIdx.x * 1024 + c] = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
}
// out[c] = sum over chunks of part[chunk][c], fixed order: 8 columns x 32 interleaved chunk groups per workgroup, four chunks of
// a group in flight (a thread's loads are each a round trip to L2: one after the other, 1024 chunks took 40 us)
__global__ __launch_bounds__(256) void k_part_final(const float* __restrict__ part, int nchunks, int cols, int pitch, float* __restrict__ out) {
  __shared__ float red[32][8];
  const int c = blockIdx.x * 8 + (threadIdx.x & 7), gq = threadIdx.x >> 3;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  if (c < cols) {
    int i = gq;
    for (; i + 96 < nchunks; i += 128) {
      a0 += part[(int64_t)i * pitch + c];
      a1 += part[(int64_t)(i + 32) * pitch + c];
      a2 += part[(int64_t)(i + 64) * pitch + c];
      a3 += part[(int64_t)(i + 96) * pitch + c];
    }
    for (; i < nchunks; i += 32) a0 += part[(int64_t)i * pitch + c];
  }
  red[gq][threadIdx.x & 7] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (gq == 0 && c < cols) {
    float a = red[0][threadIdx.x];
#pragma unroll
    for (int k = 1; k < 32; k++) a += red[k][threadIdx.x];
    out[c] = a;
  }
}
// column sums of a bf16 matrix (bias gradients): part[chunk][c] over 512-row chunks
__global__ __launch_bounds__(256) void k_colsum_bf16(const bf16* __restrict__ X, int64_t ld, int64_t rows, int cols, float* __restrict__ part) {
  __shared__ float red[8][256];
  const int c0 = blockIdx.x * 256 + (threadIdx.x & 31) * 8, gq = threadIdx.x >> 5;   // 32 lanes x 8 columns, 8 row groups
  const int64_t r0 = (int64_t)blockIdx.y * CH, r1 = r0 + CH < rows ? r0 + CH : rows;
  float acc[8];
#pragma unroll
  for (int i = 0; i < 8; i++) acc[i] = 0.f;
  if (c0 < cols)
    for (int64_t i = r0 + gq; i < r1; i += 8) {
      const bf16x8 v = *reinterpret_cast<const bf16x8*>(X + i * ld + c0);
#pragma unroll
      for (int e = 0; e < 8; e++) acc[e] += (float)v[e];
    }
#pragma unroll
  for (int e = 0; e < 8; e++) red[gq][(threadIdx.x & 31) * 8 + e] = acc[e];
  __syncthreads();
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c < cols) {
    float a = 0.f;
#pragma unroll
    for (int k = 0; k < 8; k++) a += red[k][threadIdx.x];
    part[(int64_t)blockIdx.y * cols + c] = a;
  }
}
static int colsum_bf16(hipStream_t s, const bf16* X, int64_t ld, int64_t rows, int cols, float* out, float* part) {
  const int nch = (int)((rows + CH - 1) / CH);
  hipLaunchKernelGGL(k_colsum_bf16, dim3((cols + 255) / 256, nch), dim3(256), 0, s, X, ld, rows, cols, part);
  hipLaunchKernelGGL(k_part_final, dim3((cols + 7) / 8), dim3(256), 0, s, part, nch, cols, cols, out);
  return check_launch();
}

// token-level backward of the pooling on bf16 rows: g_p = d logit_p, dx_p = (w_p / S) dxs + g_p wpool; pad rows get zeros
__global__ __launch_bounds__(256) void k_pool_bwd_tok_bf16(const bf16* __restrict__ x, const float* __restrict__ w, const float* __restrict__ S,
                                                           const float* __restrict__ xs, const float* __restrict__ dxs, const float* __restrict__ wpool,
                                                           bf16* __restrict__ dx, float* __restrict__ gout, int64_t N, int64_t Npad, int64_t P) {
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= Npad) return;
  const int lane = threadIdx.x & 63;
  bf16x8 out;
  if (row >= N) {
#pragma unroll
    for (int i = 0; i < 8; i++) out[i] = (bf16)0.f;
    *reinterpret_cast<bf16x8*>(dx + row * D + lane * 8) = out;
    return;
  }
  const int64_t b = row / P;
  const bf16x8 xv = *reinterpret_cast<const bf16x8*>(x + row * D + lane * 8);
  float a = 0.f, c0 = 0.f, dxv[8];
#pragma unroll
  for (int i = 0; i < 8; i++) {
    dxv[i] = dxs[b * D + lane * 8 + i];
    a = fmaf((float)xv[i], dxv[i], a);
    c0 = fmaf(xs[b * D + lane * 8 + i], dxv[i], c0);
  }
  a = wave_sum(a);
  c0 = wave_sum(c0);
  const float Sv = S[b], Sc = fmaxf(Sv, 1e-6f), wv = w[row];
  const float e = (a - (Sv >= 1e-6f ? c0 : 0.f)) / Sc;
  const float gv = e * wv * (1.f - wv);
#pragma unroll
  for (int i = 0; i < 8; i++) out[i] = (bf16)((wv / Sc) * dxv[i] + gv * wpool[lane * 8 + i]);
  *reinterpret_cast<bf16x8*>(dx + row * D + lane * 8) = out;
  if (lane == 0) gout[row] = gv;
}
// sum_p g_p x_p (d wpool) and sum_p g_p (d bpool): the forward's partial-sum kernel with g as the weights -- declared here
__global__ __launch_bounds__(256) void k_wsum_part_bf16(const bf16* __restrict__ x, const float* __restrict__ w, float* __restrict__ part, int64_t P) {
  __shared__ float red[4][64];
  __shared__ float sred[4];
  const int b = blockIdx.y, sl = blockIdx.z, c = blockIdx.x * 64 + (threadIdx.x & 63), gq = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int nsl = gridDim.z;
  const int64_t p0 = (int64_t)sl * PSLICE, p1 = p0 + PSLICE < P ? p0 + PSLICE : P;
  const bf16* xb = x + (int64_t)b * P * D;
  const float* wb = w + (int64_t)b * P;
  float acc = 0.f, sw = 0.f;
  for (int64_t p = p0 + gq; p < p1; p += 4) {
    const float wv = wb[p];
    sw += wv;
    acc = fmaf(wv, (float)xb[p * D + c], acc);
  }
  red[gq][lane] = acc;
  if (lane == 0) sred[gq] = sw;
  __syncthreads();
  if (gq == 0) {
    float* out = part + ((int64_t)b * nsl + sl) * (D + 1);
    out[c] = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
    if (blockIdx.x == 0 && lane == 0) out[D] = (sred[0] + sred[1]) + (sred[2] + sred[3]);
  }
}
// embedding backward, per 64-row chunk: with dpre = ds cos(pre), part[chunk][0..2][j] = sum dpre x_k, part[chunk][3][j] = sum dpre
constexpr int ECH = 64;
__global__ __launch_bounds__(256) void k_embed_bwd_part(const bf16* __restrict__ ds, const float* __restrict__ pre, const float* __restrict__ x,
                                                        int64_t N, float* __restrict__ part) {
  const int j = threadIdx.x;
  const int64_t r0 = (int64_t)blockIdx.x * ECH, r1 = r0 + ECH < N ? r0 + ECH : N;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  for (int64_t n = r0; n < r1; n++) {
    const float v = (float)ds[n * D2 + j] * cosf(pre[n * D2 + j]);
    a0 = fmaf(v, x[n * 3], a0);
    a1 = fmaf(v, x[n * 3 + 1], a1);
    a2 = fmaf(v, x[n * 3 + 2], a2);
    a3 += v;
  }
  float* out = part + (int64_t)blockIdx.x * 1024;
  out[j] = a0; out[256 + j] = a1; out[512 + j] = a2; out[768 + j] = a3;
}
// sums[k][j] (the k_part_final reduction of the chunks) -> positional.weight's [256][3] layout and positional.bias
__global__ __launch_bounds__(256) void k_embed_bwd_final(const float* __restrict__ sums, float* __restrict__ dwp, float* __restrict__ dbp) {
  const int j = threadIdx.x;
  dwp[j * 3] = sums[j]; dwp[j * 3 + 1] = sums[256 + j]; dwp[j * 3 + 2] = sums[512 + j];
  dbp[j] = sums[768 + j];
}

// ------------------------------------------------------------------------------------------------ scratch and the plan
struct BwdB {
  bf16 *dA, *dB, *dF, *dqkv, *dO, *ds, *dM;
  float *delta, *lse2, *g, *dpooled, *dxs, *slab, *part, *zeros;
  size_t bytes;
};
inline BwdB carve_bwd_b(const Shape& s, void* mem) {
  BwdB b;
  Carve c(mem);
  const size_t Np = (size_t)padded_rows(s), N = (size_t)s.N();
  b.dA = c.take<bf16>(Np * D);
  b.dB = c.take<bf16>(Np * D);
  b.dF = c.take<bf16>(Np * FF);
  b.dqkv = c.take<bf16>(Np * 3 * D);
  b.dO = c.take<bf16>(Np * D);
  b.ds = c.take<bf16>(Np * D2);
  b.dM = c.take<bf16>(Np * D);   // a block's output gradient behind its dropout (training with dropout)
  b.delta = c.take<float>(N * HEADS);
  b.lse2 = c.take<float>(N * HEADS);
  b.g = c.take<float>(N);
  b.dpooled = c.take<float>((size_t)s.B * D);
  b.dxs = c.take<float>((size_t)s.B * D);
  b.slab = c.take<float>((size_t)(SLAB_FLOATS + BIAS_PART_FLOATS));
  size_t part = colsum_part_floats(s.N(), FF);                                        // column sums, <= 2048 wide
  const size_t lnp = (size_t)((Np + lnb_rows(Np) - 1) / lnb_rows(Np)) * 1024;       // LayerNorm / embedding partials
  const size_t pool = (size_t)s.B * ((s.P + PSLICE - 1) / PSLICE) * (D + 1);
  if (lnp > part) part = lnp;
  if (pool > part) part = pool;
  b.part = c.take<float>(part);
  b.zeros = c.take<float>(FF);
  b.bytes = c.off;
  return b;
}
// workspace = [bf16 image of the parameters][its per-matrix transposes][inference activations | backward scratch]
size_t bf16_workspace_bytes(const Shape& s) {
  const size_t f = carve_b(s, nullptr, false).bytes, b = carve_bwd_b(s, nullptr).bytes;
  return 2 * wimg_bytes(s) + (f > b ? f : b);
}

#define TRY(expr)                 \
  do {                            \
    int rc__ = (expr);            \
    if (rc__) return rc__;        \
  } while (0)
inline unsigned blocks_for(int64_t n, int per) { return (unsigned)((n + per - 1) / per); }

int backward_bf16(hipStream_t s, const Shape& sh, const float* prm, const float* x, const int64_t*, const float* dout, float* dprm,
                  const void* stash, void* workspace, const Drop& dr) {
  const ParamOff po = param_offsets(sh);
  const int64_t N = sh.N(), Np = padded_rows(sh), P = sh.P;
  const int Bn = (int)sh.B;
  char* ws = reinterpret_cast<char*>(workspace);
  bf16* wT = reinterpret_cast<bf16*>(ws + wimg_bytes(sh));
  const ActsB a = carve_b(sh, const_cast<void*>(stash), true);
  const BwdB w = carve_bwd_b(sh, ws + 2 * wimg_bytes(sh));
  hipError_t e = hipMemsetAsync(w.zeros, 0, FF * sizeof(float), s);
  if (e != hipSuccess) return (int)e;
  // transposed bf16 images of the weights the dX products need
  if (4 * sh.L + 1 > 17) return SO3X_ERR_UNSUPPORTED;   // (the bf16 form is the 4-layer aircraft network)
  {
    TposeTab tab;
    tab.n = 0;
    int blocks = 0;
    auto add = [&](int64_t off, int R, int C) {
      tab.tile0[tab.n] = blocks;
      tab.off[tab.n] = off; tab.R[tab.n] = R; tab.C[tab.n] = C;
      blocks += (R / 32) * (C / 32);
      tab.n++;
    };
    for (int l = 0; l < sh.L; l++) {
      const LayerOff lo = po.layer(l);
      add(lo.wqkv, 3 * D, D); add(lo.wo, D, D); add(lo.w1, FF, D); add(lo.w2, D, FF);
    }
    add(po.wps, D2, D2);
    tab.tile0[tab.n] = blocks;
    hipLaunchKernelGGL(k_cvt_bf16_t, dim3((unsigned)blocks), dim3(256), 0, s, prm, wT, tab);
    TRY(check_launch());
  }
  const bf16* enc = a.h[sh.L];
  // head and pooling (fp32, a few rows)
  TRY(gemm(s, transposed(dout, 3), rowmajor(a.pooled, D), dprm + po.wout, D, 3, D, Bn));
  TRY(colsum(s, dout, 3, Bn, 3, dprm + po.bout, w.part));
  TRY(gemm(s, rowmajor(dout, 3), rowmajor(prm + po.wout, D), w.dpooled, D, Bn, D, 3));
  TRY(gemm(s, transposed(w.dpooled, D), rowmajor(a.xs, D), dprm + po.wlin, D, D, D, Bn));
  TRY(colsum(s, w.dpooled, D, Bn, D, dprm + po.blin, w.part));
  TRY(gemm(s, rowmajor(w.dpooled, D), rowmajor(prm + po.wlin, D), w.dxs, D, Bn, D, D));
  hipLaunchKernelGGL(k_pool_bwd_tok_bf16, dim3(blocks_for(Np, 4)), dim3(256), 0, s, enc, a.w, a.S, a.xs, w.dxs, prm + po.wpool, w.dA, w.g, N, Np, P);
  const int nsl = (int)((P + PSLICE - 1) / PSLICE);
  hipLaunchKernelGGL(k_wsum_part_bf16, dim3(D / 64, (unsigned)sh.B, nsl), dim3(256), 0, s, enc, w.g, w.part, P);
  hipLaunchKernelGGL(k_part_final, dim3((D + 1 + 7) / 8), dim3(256), 0, s, w.part, Bn * nsl, D + 1, D + 1, dprm + po.wpool);   // wpool[512] | bpool[1]
  TRY(check_launch());
  const float sc = 1.f / sqrtf((float)DH), c2 = sc * 1.4426950408889634f;
  const int LNB_ROWS = lnb_rows(Np), lnblk = (int)((Np + LNB_ROWS - 1) / LNB_ROWS);
  bf16* dcur = w.dA;
  bf16* dalt = w.dB;
  for (int l = sh.L - 1; l >= 0; l--) {
    const LayerOff lo = po.layer(l);
    const LayerB& k = a.layer[l];
    const bf16* h = a.h[l];
    // norm2: dcur = d h_{l+1} -> dalt = d r2;  d gamma2 | d beta2 are adjacent in the parameter buffer
    bf16* const dM = dr.on() ? w.dM : nullptr;
    hipLaunchKernelGGL(k_ln_bwd_bf16, dim3(lnblk), dim3(256), 0, s, dcur, k.r2, k.st2, prm + lo.g2, dalt, w.part, Np, dM, gemm_drop(dr, l, DROP_BLOCK2), LNB_ROWS);
    hipLaunchKernelGGL(k_part_final, dim3(128), dim3(256), 0, s, w.part, lnblk, 1024, 1024, dprm + lo.g2);
    TRY(check_launch());
    // feed-forward: r2 = x1 + [dropout] (relu(x1 W1^T + b1) [dropout]) W2^T + b2; dy2 = the gradient behind the output dropout
    const bf16* dy2 = dr.on() ? w.dM : dalt;
    GemmDrop relu_scale;   // the hidden activations' dropout: f is stored dropped out, so (f > 0) is "active and kept"
    relu_scale.scale = dr.on() ? dr.inv_keep() : 1.f;
    TRY(gemm_tn(s, dy2, D, k.f, FF, dprm + lo.w2, (int)Np, D, FF, w.slab, dprm + lo.b2));   // (+ d b2: rows >= N of every dY are zero)
    TRY(gemm_bf16(s, dy2, D, wT + lo.w2, D, w.dF, FF, w.zeros, k.f, FF, (int)Np, FF, D, EPI_MASK, relu_scale));   // dZ = (dy2 W2) o (f > 0)
    TRY(gemm_tn(s, w.dF, FF, k.x1, D, dprm + lo.w1, (int)Np, FF, D, w.slab, dprm + lo.b1));
    TRY(gemm_bf16(s, w.dF, FF, wT + lo.w1, FF, dcur, D, w.zeros, dalt, D, (int)Np, D, FF, EPI_RESID));        // dcur = d x1 = dr2 + dZ W1
    // norm1: dcur = d x1 -> dalt = d r1
    hipLaunchKernelGGL(k_ln_bwd_bf16, dim3(lnblk), dim3(256), 0, s, dcur, k.r1, k.st1, prm + lo.g1, dalt, w.part, Np, dM, gemm_drop(dr, l, DROP_BLOCK1), LNB_ROWS);
    hipLaunchKernelGGL(k_part_final, dim3(128), dim3(256), 0, s, w.part, lnblk, 1024, 1024, dprm + lo.g1);
    TRY(check_launch());
    // attention block: r1 = h + [dropout] (softmax(Q K^T / sqrt(dh)) [dropout] V Wo^T + bo)
    const bf16* dy1 = dr.on() ? w.dM : dalt;
    TRY(gemm_tn(s, dy1, D, k.o, D, dprm + lo.wo, (int)Np, D, D, w.slab, dprm + lo.bo));
    TRY(gemm_bf16(s, dy1, D, wT + lo.wo, D, w.dO, D, w.zeros, nullptr, 0, (int)Np, D, D, EPI_NONE));
    hipLaunchKernelGGL(k_attn_delta, dim3(blocks_for(N * HEADS, 16)), dim3(256), 0, s, k.o, w.dO, k.lse, w.delta, w.lse2, N, (int)P);
    const dim3 ag((unsigned)((P + 127) / 128 * HEADS * sh.B));
    if (dr.on()) {
      const float ik = dr.inv_keep();
      hipLaunchKernelGGL(k_attn_bwd_dq<true>, ag, dim3(256), 0, s, k.qkv, w.dO, w.lse2, w.delta, w.dqkv, (int)P, sc, c2, k.maskq, ik);
      hipLaunchKernelGGL((k_attn_bwd_dkv<0, true>), ag, dim3(256), 0, s, k.qkv, w.dO, w.lse2, w.delta, w.dqkv, (int)P, sc, c2, k.maskk, ik);
      hipLaunchKernelGGL((k_attn_bwd_dkv<1, true>), ag, dim3(256), 0, s, k.qkv, w.dO, w.lse2, w.delta, w.dqkv, (int)P, sc, c2, k.maskk, ik);
    } else {
      const uint32_t* nm = nullptr;
      hipLaunchKernelGGL(k_attn_bwd_dq<false>, ag, dim3(256), 0, s, k.qkv, w.dO, w.lse2, w.delta, w.dqkv, (int)P, sc, c2, nm, 1.f);
      hipLaunchKernelGGL((k_attn_bwd_dkv<0, false>), ag, dim3(256), 0, s, k.qkv, w.dO, w.lse2, w.delta, w.dqkv, (int)P, sc, c2, nm, 1.f);
      hipLaunchKernelGGL((k_attn_bwd_dkv<1, false>), ag, dim3(256), 0, s, k.qkv, w.dO, w.lse2, w.delta, w.dqkv, (int)P, sc, c2, nm, 1.f);
    }
    TRY(check_launch());
    if (Np > N) {
      e = hipMemsetAsync(w.dqkv + N * 3 * D, 0, (size_t)(Np - N) * 3 * D * sizeof(bf16), s);
      if (e != hipSuccess) return (int)e;
    }
    TRY(gemm_tn(s, w.dqkv, 3 * D, h, D, dprm + lo.wqkv, (int)Np, 3 * D, D, w.slab, dprm + lo.bqkv));
    TRY(gemm_bf16(s, w.dqkv, 3 * D, wT + lo.wqkv, 3 * D, dcur, D, w.zeros, dalt, D, (int)Np, D, 3 * D, EPI_RESID));   // dcur = d h = dr1 + dqkv Wqkv
  }
  // embedding: h0[:, :256] = sin(pre) Wps^T + bps, pre = x Wp^T + bp
  TRY(gemm_tn(s, dcur, D, a.sn, D2, dprm + po.wps, (int)Np, D2, D2, w.slab));
  TRY(colsum_bf16(s, dcur, D, N, D2, dprm + po.bps, w.part));
  TRY(gemm_bf16(s, dcur, D, wT + po.wps, D2, w.ds, D2, w.zeros, nullptr, 0, (int)Np, D2, D2, EPI_NONE));
  const int nch = (int)((N + ECH - 1) / ECH);
  hipLaunchKernelGGL(k_embed_bwd_part, dim3(nch), dim3(256), 0, s, w.ds, a.pre, x, N, w.part);
  hipLaunchKernelGGL(k_part_final, dim3(128), dim3(256), 0, s, w.part, nch, 1024, 1024, w.slab);
  hipLaunchKernelGGL(k_embed_bwd_final, dim3(1), dim3(256), 0, s, w.slab, dprm + po.wp, dprm + po.bp);
  return check_launch();
}

}  // namespace plane
}  // namespace so3x
