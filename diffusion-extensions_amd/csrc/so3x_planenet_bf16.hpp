// so3x_planenet_bf16.hpp -- types, LDS-DMA helper, activation buffers and kernel launchers shared by the forward
// (so3x_planenet_bf16.hip) and the backward (so3x_planenet_bf16_bwd.hip) of the bf16 form of PlaneNet.
#pragma once
#include "so3x_planenet.hpp"
#include "so3x_math.hpp"

namespace so3x {
namespace plane {

typedef __bf16 bf16;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

#define GLDS16(SRC, DST)                                                                                            \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(SRC),                            \
                                   (__attribute__((address_space(3))) void*)(DST), 16, 0, 0)

// LDS-DMA the compiler does not know about: 16 (or 4) bytes per lane from `src` (per lane) to LDS `dst + 16 (4) lane` (dst
// wave-uniform).  The builtin form makes every LDS read whose address the compiler cannot bound -- the ds_read_b64_tr_b16
// intrinsic's, i.e. every transposed operand -- wait for `vmcnt(0)`: for the tile requested a few hundred cycles earlier, which
// takes the prefetch out of a double-buffered loop.  Ordering is the kernel's own: one `s_waitcnt vmcnt(0)` + barrier between a
// tile's DMA and its first read.
__device__ __forceinline__ void glds16_asm(const void* src, const void* dst) {
  typedef __attribute__((address_space(3))) const char* lds_cp;
  const uint32_t d = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(lds_cp)dst);
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(src), "s"(d) : "memory", "m0");
}
__device__ __forceinline__ void glds4_asm(const void* src, const void* dst) {
  typedef __attribute__((address_space(3))) const char* lds_cp;
  const uint32_t d = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(lds_cp)dst);
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" :: "v"(src), "s"(d) : "memory", "m0");
}

// keep bits of the eight elements 8 c .. 8 c + 7 of a dropout site: bit j = piece j (words x, y, z, w; low half first) >= thr16
__device__ __forceinline__ uint32_t drop_keep8(uint64_t seed, uint64_t c, uint64_t ctr_hi, uint32_t thr16) {
  const Philox4 r = philox4x32_10(seed, c, ctr_hi);
  const uint32_t u[4] = {r.x, r.y, r.z, r.w};
  uint32_t m = 0;
#pragma unroll
  for (int j = 0; j < 8; j++) m |= (((u[j >> 1] >> (16 * (j & 1))) & 0xFFFFu) >= thr16 ? 1u : 0u) << j;
  return m;
}

constexpr int D = 512, HEADS = 4, DH = 128, FF = 2048, D2 = 256;

// The attention kernels' workgroup -> (tile xt of the query / key axis, head, cloud) map.  All `nxt` workgroups of one (cloud, head)
// stream the same K / V (or Q / dO) rows; workgroup ids go round-robin over the 8 XCDs, so in the natural (x, y, z) order each of
// them ran on a different XCD and every XCD's L2 fetched every (cloud, head) for itself (PMC: 1.09 GB fetched per forward launch at
// 32 x 2048 against 0.2 GB of operands).  Here XCD x owns a contiguous range of the (cloud, head, tile) order (bijective for any
// grid size): the tiles of a (cloud, head) follow each other on ONE XCD and find its rows in that L2.  Grid: 1-D, nxt * HEADS * B.
__device__ __forceinline__ void attn_tile(int nxt, int& xt, int& hd, int& b) {
  const int total = gridDim.x, bid = blockIdx.x, xcd = bid & 7, qq = total >> 3, rr = total & 7;
  const int pos = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (bid >> 3);
  xt = pos % nxt;
  const int bh = pos / nxt;
  hd = bh % HEADS;
  b = bh / HEADS;
}

// LDS image of a [64 keys][128] bf16 tile: 256-byte rows, the row's 16-byte chunk c at position c ^ swz16(row) -- conflict-free for
// the row reads of K (ds_read_b128) and for the transposed reads of V (ds_read_b64_tr_b16)
__device__ __forceinline__ int swz16(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }

// sum over the 8 lanes that share lane >> 3 (every one of them gets it), on the vector ALU's data-parallel primitives: quad_perm
// [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror -- three full-rate instructions where __shfl_xor is three LDS round trips
__device__ __forceinline__ float sum8(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));    // quad_perm 1,0,3,2
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));    // quad_perm 2,3,0,1
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));   // row_half_mirror
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

constexpr int PSLICE = 64;    // points per partial sum of the pooling reductions (32 clouds x 256 points: 128 workgroups, not 32)

// ------------------------------------------------------------------------------------------------ buffers
inline int64_t padded_rows(const Shape& s) { return (s.N() + 127) / 128 * 128; }

// maskq / maskk (training stash only): the attention dropout's keep bits of a layer, [B][H][P][P / 32] words -- bit (k & 31) of word
// k >> 5 in row q of maskq, and the transpose (bit of q in row k) in maskk: the kernels with the query on the lane read the
// one, those with the key on the lane the other (written by k_attn_mask when a forward runs with dropout)
struct LayerB { bf16 *qkv, *o, *r1, *x1, *f, *r2; float *st1, *st2, *lse; uint32_t *maskq, *maskk; };
struct ActsB {
  float* pre;              // [N][256] fp32 SIREN pre-activations (kept for the backward only)
  bf16* sn;                // [Npad][256]
  bf16* h[66];
  LayerB layer[65];
  float *w, *S, *xs, *pooled, *part, *temb;
  float* stat_part;        // [Npad][8][2] row sums / sums of squares of a residual product's output (LayerNorm-folded inference)
  size_t bytes;
};
inline ActsB carve_b(const Shape& s, void* mem, bool per_layer) {
  ActsB a;
  Carve c(mem);
  const size_t Np = (size_t)padded_rows(s);
  a.pre = per_layer ? c.take<float>((size_t)s.N() * D2) : nullptr;
  a.sn = c.take<bf16>(Np * D2);
  if (per_layer) {
    for (int l = 0; l <= s.L; l++) a.h[l] = c.take<bf16>(Np * D);
  } else {
    bf16* h0 = c.take<bf16>(Np * D);
    bf16* h1 = c.take<bf16>(Np * D);
    for (int l = 0; l <= s.L; l++) a.h[l] = (l & 1) ? h1 : h0;
  }
  for (int l = 0; l < s.L; l++) {
    if (l == 0 || per_layer) {
      LayerB& k = a.layer[l];
      k.qkv = c.take<bf16>(Np * 3 * D);
      k.o = c.take<bf16>(Np * D);
      k.r1 = c.take<bf16>(Np * D);
      k.x1 = c.take<bf16>(Np * D);
      k.f = c.take<bf16>(Np * FF);
      k.r2 = c.take<bf16>(Np * D);
      k.st1 = c.take<float>(Np * 2);
      k.st2 = c.take<float>(Np * 2);
      k.lse = c.take<float>((size_t)s.N() * HEADS);
      k.maskq = k.maskk = nullptr;
      if (per_layer) {
        k.maskq = c.take<uint32_t>((size_t)s.B * HEADS * s.P * (s.P / 32));
        k.maskk = c.take<uint32_t>((size_t)s.B * HEADS * s.P * (s.P / 32));
      }
    } else {
      a.layer[l] = a.layer[0];
    }
  }
  a.w = c.take<float>((size_t)s.N());
  a.S = c.take<float>((size_t)s.B);
  a.xs = c.take<float>((size_t)s.B * D);
  a.pooled = c.take<float>((size_t)s.B * D);
  a.part = c.take<float>((size_t)s.B * ((s.P + PSLICE - 1) / PSLICE) * (D + 1));
  a.temb = c.take<float>((size_t)s.B * D2);
  a.stat_part = c.take<float>(Np * 16);
  a.bytes = c.off;
  return a;
}

// workspace = [bf16 image of the parameters][inference activations | backward scratch]
// The weight image: the plain bf16 copy of the parameters, then what the LayerNorm-folded inference path reads (GemmLN): per layer
// W1' = W1 gamma1 [F][D] and Wqkv' = Wqkv gamma2 of the PREVIOUS layer [3 D][D] as bf16 (layer 0's slot unused), then the fp32
// vectors s1[F], c1[F], sq[3 D], cq[3 D] (s = row sums of the bf16 W', c = W beta + bias).
inline size_t wimg_plain_bytes(const Shape& s) { return up((size_t)param_offsets(s).total * sizeof(bf16)); }
struct FoldOff { size_t w1, wqkv, s1, c1, sq, cq; };   // byte offsets from the image base
inline FoldOff fold_off(const Shape& s, int l) {
  const size_t mats = ((size_t)FF * D + (size_t)3 * D * D) * sizeof(bf16), vecs = ((size_t)2 * FF + 6 * D) * sizeof(float);
  FoldOff o;
  o.w1 = wimg_plain_bytes(s) + (size_t)l * mats;
  o.wqkv = o.w1 + (size_t)FF * D * sizeof(bf16);
  o.s1 = wimg_plain_bytes(s) + (size_t)s.L * mats + (size_t)l * vecs;
  o.c1 = o.s1 + FF * sizeof(float);
  o.sq = o.c1 + FF * sizeof(float);
  o.cq = o.sq + 3 * D * sizeof(float);
  return o;
}
inline size_t wimg_bytes(const Shape& s) {
  return up(wimg_plain_bytes(s) + (size_t)s.L * (((size_t)FF * D + (size_t)3 * D * D) * sizeof(bf16) + ((size_t)2 * FF + 6 * D) * sizeof(float)));
}


// C[M][N] = epilogue(A[M][K] W[N][K]^T + bias[N]),  M % 128 == N % 128 == K % 64 == 0; every operand bf16 row-major, bias fp32
//   EPI_NONE: C = acc + bias;  EPI_RELU: max(., 0);  EPI_RESID: + R[M][N];  EPI_MASK: (R[M][N] > 0) ? acc + bias : 0
enum Epi { EPI_NONE = 0, EPI_RELU = 1, EPI_RESID = 2, EPI_MASK = 3 };
// Training-mode dropout inside an epilogue (Drop, so3x_planenet.hpp; the element index is row * N + column of the LOGICAL [M][N]
// matrix): EPI_RELU and EPI_RESID drop (and rescale by `scale` = 1 / keep) the value acc + bias [after the ReLU] -- EPI_RESID before
// the residual is added --, EPI_MASK multiplies what passes by `scale`.  thr16 == 0 and scale == 1: the plain epilogues.
struct GemmDrop {
  uint32_t thr16 = 0;
  float scale = 1.f;
  uint64_t seed = 0, ctr_hi = 0;
};
inline GemmDrop gemm_drop(const Drop& dr, int layer, int site) {
  return dr.on() ? GemmDrop{dr.thr16(), dr.inv_keep(), dr.seed, dr.ctr_hi(layer, site)} : GemmDrop{};
}
int gemm_bf16(hipStream_t s, const bf16* A, int lda, const bf16* W, int ldw, bf16* C, int ldc, const float* bias, const bf16* R, int ldr,
              int M, int N, int K, int epi, GemmDrop gd = GemmDrop{});

// LayerNorm folded into the products around it (inference, large token counts: the persistent 256-wide kernel only).  With
// r the un-normalised row, (mean, rstd) its statistics, y = (r - mean) rstd gamma + beta:
//   LN_A     the product's A operand is r instead of y: y W^T = rstd (r W'^T) - rstd mean s + c  with  W' = W gamma (per column),
//            s = W' 1, c = W beta + bias -- a row scale and a rank-one correction in the epilogue (`bias` = c, `svec` = s, `rowstat`);
//   LN_STAT  (EPI_RESID) the output rows' sum and sum of squares over this wave's 64 columns go to stat_part[row][8][2] (a tiny
//            kernel turns the eight partials into (mean, rstd));
//   LN_RESID LN_STAT + the residual operand is r: its y is recomputed in the epilogue from rowstat, gamma, beta.
// The normalised rows are then never written or read: k_ln_bf16 -- 128 MB of HBM traffic per LayerNorm at 32 x 2048 points --
// drops out of the forward (section 4 of DESIGN.md).
enum { LN_NONE = 0, LN_A = 1, LN_STAT = 2, LN_RESID = 3 };
struct GemmLN {
  const float* rowstat = nullptr;   // [M][2] (mean, rstd) of the rows of A (LN_A) or of R (LN_RESID)
  const float* svec = nullptr;      // [N]  LN_A
  const float* gamma = nullptr;     // [N]  LN_RESID
  const float* beta = nullptr;      // [N]  LN_RESID
  float* stat_part = nullptr;       // [M][8][2]  LN_STAT, LN_RESID
};
bool gemm_bf16_ln_ok(int M, int N, int K);
int gemm_bf16_ln(hipStream_t s, const bf16* A, int lda, const bf16* W, int ldw, bf16* C, int ldc, const float* bias, const bf16* R, int ldr,
                 int M, int N, int K, int epi, int lnm, GemmLN ln);

}  // namespace plane
}  // namespace so3x
