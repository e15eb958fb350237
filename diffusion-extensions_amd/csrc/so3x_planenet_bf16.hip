// so3x_planenet_bf16.hip -- the bf16 matrix-core form of the PlaneNet denoiser (reference models.py:185-210) at the aircraft
// task's own shape: dim 512, 4 heads of 128, feed-forward 2048, points a multiple of 64 (aircraft_rotate.py:17-47).
//
// Data layout in HBM: activations token-major [tokens][width] in bf16 (tokens padded up to a multiple of 128 with zero rows;
// rows never mix outside attention and pooling, which index real tokens only); LayerNorm statistics, attention log-sum-exp,
// pooling weights and everything behind the pooling in fp32; the fp32 master parameters are converted to a bf16 image once per
// call.  Kernels:
//   k_gemm_bf16      C = A W^T (+ bias, ReLU, + residual) -- 128 x 128 x 64 tiles, 4 waves x (4 x 4) v_mfma_f32_16x16x32_bf16,
//                    both operands K-contiguous, staged by LDS-DMA (global_load_lds_dwordx4) into two XOR-swizzled LDS buffers
//                    (swizzle on the SOURCE address, conflict-free ds_read_b128), next tile in flight under the current one,
//                    fp32 accumulators staged through LDS for whole-row stores, XCD-contiguous tile order.  MFMA-bound.
//   k_attn_fwd       softmax(Q K^T / sqrt(128)) V per (cloud, head), flash style: 4 waves x 32 queries, 64-key tiles of K and V
//                    by LDS-DMA (double-buffered), S^T = K Q^T on v_mfma_f32_32x32x16_bf16 so that a query's scores sit in one
//                    lane pair (softmax without LDS), P^T fed to O^T = V^T P^T straight from the accumulator registers, V^T
//                    operands by ds_read_b64_tr_b16.  Scores never touch HBM.  MFMA / VALU(exp) bound.
//   k_ln_bf16        LayerNorm rows (HBM-bound: 2 B in + 2 B out per element).
//   k_embed_bf16, k_pool_* : the SIREN / time embedding and PoolRN's weighted mean (HBM-bound).
#include <type_traits>

#include "so3x_planenet_bf16.hpp"

namespace so3x {
namespace plane {

// (layers <= 4: the backward's transpose table holds 4 layers' matrices + the head -- refused here, before a forward builds a stash)
bool bf16_supported(const Shape& s) { return s.d == D && s.H == HEADS && s.F == FF && s.P % 64 == 0 && s.P >= 64 && s.L <= 4; }

// ------------------------------------------------------------------------------------------------ fp32 -> bf16 image
__global__ __launch_bounds__(256) void k_cvt_bf16(const float* __restrict__ src, bf16* __restrict__ dst, int64_t n4) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  const float4 v = reinterpret_cast<const float4*>(src)[i];
  reinterpret_cast<bf16x4*>(dst)[i] = bf16x4{(bf16)v.x, (bf16)v.y, (bf16)v.z, (bf16)v.w};
}
__global__ __launch_bounds__(256) void k_cvt_f32(const bf16* __restrict__ src, float* __restrict__ dst, int64_t n4) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  const bf16x4 v = reinterpret_cast<const bf16x4*>(src)[i];
  reinterpret_cast<float4*>(dst)[i] = float4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
}

// ------------------------------------------------------------------------------------------------ GEMM
// C[M][N] = act(A[M][K] W[N][K]^T + bias[N] (+ R[M][N])),  M % 128 == N % 128 == K % 64 == 0
constexpr int BM = 128, BN = 128, BK = 64;
// the 128-wide kernel's fp32 C tile in LDS: rows of 132 floats.  With 128 (512 bytes) the four row groups a wave's accumulator store
// touches (lane >> 4) fell into the same banks -- 4-way conflicts on all 64 stores of the epilogue, 15-22 % of the kernel's LDS cycles
// by PMC (r05); 132 puts them 16 banks apart.  Dynamic LDS (two workgroups per CU still fit).
constexpr int CLD = 132;
// dynamic LDS of the MI-fragment form: the larger of its two staging buffers and its C tile (MI = 4: 67,584 bytes; MI = 2: 49,152)
// (three stages for the short tile -- two K tiles in flight -- were measured: forward 425 -> 429 us at 32 x 256, not kept)
template <int MI> constexpr int gemm128_stages() { return 2; }
template <int MI> constexpr int gemm128_lds() {
  return gemm128_stages<MI>() * (32 * MI * 128 + 16384) > 32 * MI * CLD * 4 ? gemm128_stages<MI>() * (32 * MI * 128 + 16384) : 32 * MI * CLD * 4;
}
static PerDevice g_gemm128[16];

// MI = 16-row fragments per wave along M: 4 -> a 128 x 128 tile, 2 -> a 64 x 128 tile (48 KB of staging + C).  The short tile is for
// the products whose 128-row tiling would not fill the chip twice over: at the 8,192 tokens the reference trains PlaneNet with (32 x
// 256, aircraft_rotate.py:17-30) a 512-wide output is 256 tiles -- ONE workgroup per CU, whose K tiles (one exposed load round trip
// each: `vmcnt(0)` + barrier) then have nothing to overlap with: 4.3 GFLOP in 22 us.
template <int EPI, bool DROP, int MI>
__global__ __launch_bounds__(256, 2) void k_gemm_bf16(const bf16* __restrict__ A, const bf16* __restrict__ W, bf16* __restrict__ C,
                                                      const float* __restrict__ bias, const bf16* __restrict__ R, int M, int N, int K,
                                                      int lda, int ldw, int ldc, int ldr, const GemmDrop gd) {
  constexpr int TM = 32 * MI, ABYTES = TM * 128, STAGE = ABYTES + 16384;   // A tile | W tile (16 KB)
  constexpr int NST = gemm128_stages<MI>();                     // K tiles in LDS: the one being read + NST - 1 in flight
  extern __shared__ __attribute__((aligned(16))) char smem[];   // NST x (A tile | W tile); then the fp32 C tile, rows of CLD floats
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
  const int ntn = N / BN, nwg = ntn * (M / TM);
  // XCD-contiguous tile order (workgroup ids round-robin over the 8 XCDs; bijective for any nwg): tiles that share an A panel
  // run on one XCD and find it in that XCD's L2
  const int bid = blockIdx.x, xcd = bid & 7, qq = nwg >> 3, rr = nwg & 7;
  const int tile = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (bid >> 3);
  const int tm = tile / ntn, tn = tile % ntn;
  const bf16* Ag = A + (size_t)tm * TM * lda;
  const bf16* Wg = W + (size_t)tn * BN * ldw;
  // staging: wave w moves rows 32 w .. 32 w + 31 of the W tile and rows 8 MI w .. 8 MI (w + 1) - 1 of the A tile, 8 rows (1 KB) per
  // LDS-DMA instruction; LDS position (row, chunk c) holds the row's 16-byte chunk c ^ ((row >> 1) & 7)
  auto stage = [&](int kt, int buf) {
    char* sa = smem + buf * STAGE;
    char* sb = sa + ABYTES;
#pragma unroll
    for (int i = 0; i < MI; i++) {
      const int row = wave * 8 * MI + i * 8 + (lane >> 3);
      const int ch = (lane & 7) ^ ((row >> 1) & 7);
      GLDS16(Ag + (size_t)row * lda + kt * BK + ch * 8, sa + (wave * 8 * MI + i * 8) * 128);
    }
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int row = wave * 32 + i * 8 + (lane >> 3);
      const int ch = (lane & 7) ^ ((row >> 1) & 7);
      GLDS16(Wg + (size_t)row * ldw + kt * BK + ch * 8, sb + (wave * 32 + i * 8) * 128);
    }
  };
  f32x4 acc[MI][4];
#pragma unroll
  for (int i = 0; i < MI; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int fsw = (lane >> 1) & 7;                       // ((row >> 1) & 7) of this lane's fragment rows (row = 16 x + (lane & 15))
  const int arow = (wm * 16 * MI + (lane & 15)) * 128, brow = (wn * 64 + (lane & 15)) * 128;
  const int KT = K / BK;
  stage(0, 0);
  if (NST == 3 && KT > 1) stage(1, 1);
  int buf = 0;
  for (int kt = 0; kt < KT; kt++) {
    // this wave's pieces of tile kt have landed (vmcnt counts in issue order: with three stages the MI + 4 pieces of tile kt + 1
    // may stay in flight)
    if (NST == 3 && kt + 1 < KT) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(MI + 4) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                     // ... everyone's have, and everyone is done reading the buffer staged next
    if (kt + NST - 1 < KT) stage(kt + NST - 1, buf == 0 ? NST - 1 : buf - 1);
    const char* sa = smem + buf * STAGE;
    const char* sb = sa + ABYTES;
    buf = buf == NST - 1 ? 0 : buf + 1;
#pragma unroll
    for (int s = 0; s < 2; s++) {
      const int choff = ((4 * s + (lane >> 4)) ^ fsw) << 4;
      bf16x8 a[MI], b[4];
#pragma unroll
      for (int i = 0; i < MI; i++) a[i] = *reinterpret_cast<const bf16x8*>(sa + arow + i * 2048 + choff);
#pragma unroll
      for (int i = 0; i < 4; i++) b[i] = *reinterpret_cast<const bf16x8*>(sb + brow + i * 2048 + choff);
#pragma unroll
      for (int i = 0; i < MI; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
  }
  __syncthreads();
  float* sc = reinterpret_cast<float*>(smem);            // [TM][CLD] fp32
#pragma unroll
  for (int i = 0; i < MI; i++)
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
      for (int e = 0; e < 4; e++) sc[(wm * 16 * MI + i * 16 + (lane >> 4) * 4 + e) * CLD + wn * 64 + j * 16 + (lane & 15)] = acc[i][j][e];
  __syncthreads();
  const int c4 = (tid & 31) * 4;
  const float4 bv = *reinterpret_cast<const float4*>(bias + tn * BN + c4);
  // the residual / mask values of this thread's rows, all requested before the first C store (a load issued behind a store
  // cannot be waited for without retiring the store: vmcnt counts in issue order)
  constexpr int NIT = TM / 8;
  bf16x4 rall[NIT];
  if constexpr (EPI == EPI_RESID || EPI == EPI_MASK) {
#pragma unroll
    for (int it = 0; it < NIT; it++)
      rall[it] = *reinterpret_cast<const bf16x4*>(R + ((size_t)tm * TM + it * 8 + (tid >> 5)) * ldr + tn * BN + c4);
  }
#pragma unroll
  for (int it = 0; it < NIT; it++) {
    const int row = it * 8 + (tid >> 5);
    float4 v = *reinterpret_cast<const float4*>(sc + row * CLD + c4);
    v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
    const size_t grow = (size_t)tm * TM + row;
    if constexpr (EPI == EPI_RELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
    if constexpr (DROP && (EPI == EPI_RELU || EPI == EPI_RESID)) {
      {   // (four columns per thread: one half of a call's eight flags)
        const uint64_t e = (uint64_t)grow * N + tn * BN + c4;
        const uint32_t m = drop_keep8(gd.seed, e >> 3, gd.ctr_hi, gd.thr16) >> (c4 & 4);
        v.x = m & 1 ? v.x * gd.scale : 0.f; v.y = m & 2 ? v.y * gd.scale : 0.f;
        v.z = m & 4 ? v.z * gd.scale : 0.f; v.w = m & 8 ? v.w * gd.scale : 0.f;
      }
    }
    if constexpr (EPI == EPI_RESID || EPI == EPI_MASK) {
      const bf16x4 rv = rall[it];
      if constexpr (EPI == EPI_RESID) {
        v.x += (float)rv[0]; v.y += (float)rv[1]; v.z += (float)rv[2]; v.w += (float)rv[3];
      } else {
        const float ms = DROP ? gd.scale : 1.f;
        v.x = (float)rv[0] > 0.f ? v.x * ms : 0.f; v.y = (float)rv[1] > 0.f ? v.y * ms : 0.f;
        v.z = (float)rv[2] > 0.f ? v.z * ms : 0.f; v.w = (float)rv[3] > 0.f ? v.w * ms : 0.f;
      }
    }
    *reinterpret_cast<bf16x4*>(C + grow * ldc + tn * BN + c4) = bf16x4{(bf16)v.x, (bf16)v.y, (bf16)v.z, (bf16)v.w};
  }
}

// ---- the large-M form: 256 x 256 x 64 tiles, 8 waves (2 x 4, 128 x 64 of C each), PERSISTENT workgroups ------------------------
// One workgroup per CU walks its tiles; the K-tiles of all its tiles form ONE stream through two 64 KB LDS buffers, each made of
// four 16 KB half-tiles (A rows 0-127 | A rows 128-255 | W rows 0-127 | W rows 128-255).  A K-tile is two phases of 32 MFMAs per
// wave, each phase two halves with a raw barrier behind each:
//     PA.H1  LDS reads: A(rows 0-63 of the wave's 128), W(all 64 of its columns); LDS-DMA: both A halves of K-tile u+1
//     PA.H2  C rows 0-63 of the wave                                                  (32 MFMAs)
//     PB.H1  LDS reads: A(rows 64-127); LDS-DMA: both W halves of K-tile u+2; `s_waitcnt vmcnt(4)`
//     PB.H2  C rows 64-127
// Waves 4-7 (row half wm = 1; wave w and w + 4 share a SIMD) run ONE barrier behind waves 0-3, so while one wave of a SIMD
// computes, the other one fetches: the SIMD's matrix pipe alternates between them instead of idling while both wait for the LDS.
// The one counted wait per K-tile leaves exactly the four pieces staged last (W of u+2) in flight and retires all of K-tile u+1; it
// sits in PB.H1, so both groups have passed it, and a barrier, before either reads K-tile u+1 (a staged buffer is read one
// barrier after the wait that retires it; a half-tile is re-staged at least one barrier after its last read).  The stream does
// not stop at a tile boundary: the next tile's first K-tiles are in flight while this tile's accumulators are stored, so neither
// the prologue latency nor the store tail is exposed.  Staging addresses are a scalar base per (tile, half, K-tile) plus a
// per-lane offset fixed for the whole launch.
// C^T = W A^T is what the MFMAs compute (W fragment as the A operand), so a lane holds 4 consecutive columns of one C row: one
// ds_write_b128 per tile into the epilogue's LDS transposition (below).
constexpr int TB = 256;

template <int EPI, bool DROP, int LNM = LN_NONE>
__global__ __launch_bounds__(512, 2) void k_gemm256_bf16(const bf16* __restrict__ A, const bf16* __restrict__ W, bf16* __restrict__ C,
                                                         const float* __restrict__ bias, const bf16* __restrict__ R, int M, int N, int K,
                                                         int lda, int ldw, int ldc, int ldr, const GemmDrop gd, const GemmLN ln = GemmLN{}) {
  constexpr int NST = LNM >= LN_STAT ? 32 : 16;   // stores per wave and tile (the statistics are one more per 8-row unit)
  __shared__ __attribute__((aligned(16))) char smem[131072 + 32768];   // two K-tile buffers | 4 KB per wave for the epilogue
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), wm = wave >> 2, wn = wave & 3;
  const int ntn = N / TB, ntiles = ntn * (M / TB), KT = K / 64;
  // this workgroup's tiles: XCD x (workgroup ids round-robin over the 8 XCDs) owns a contiguous range of tiles, its workgroups take
  // them round-robin -- at any time an XCD works on neighbouring tiles, which share their A panel through that XCD's L2
  const int G = gridDim.x, bid = blockIdx.x, xcd = bid & 7, qq = ntiles >> 3, rr = ntiles & 7;
  const int x_start = xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq, x_cnt = qq + (xcd < rr ? 1 : 0);
  const int nlb = (G >> 3) + ((G & 7) > xcd ? 1 : 0), lb = bid >> 3;
  const int nmine = lb < x_cnt ? (x_cnt - lb + nlb - 1) / nlb : 0;
  if (nmine == 0) return;
  // per-lane byte offsets of the two 1 KB pieces a wave moves per half-tile (rows 16 w + 8 i + (lane >> 3); the 16-byte chunk is
  // the LDS position's chunk ^ ((row >> 1) & 7))
  unsigned voa[2], vow[2];
#pragma unroll
  for (int i = 0; i < 2; i++) {
    const int row = wave * 16 + i * 8 + (lane >> 3), ch = (lane & 7) ^ ((row >> 1) & 7);
    voa[i] = (unsigned)(row * lda + ch * 8) * 2u;
    vow[i] = (unsigned)(row * ldw + ch * 8) * 2u;
  }
  // stream cursors: the A halves are staged one K-tile ahead, the W halves two
  struct Cur { int it, kt; const char* pa; const char* pw; };
  auto tile_base = [&](Cur& c) {
    if (c.it < nmine) {
      const int tile = x_start + lb + c.it * nlb, tm = tile / ntn, tn = tile - tm * ntn;
      c.pa = reinterpret_cast<const char*>(A + (size_t)tm * TB * lda);
      c.pw = reinterpret_cast<const char*>(W + (size_t)tn * TB * ldw);
    }
  };
  auto advance = [&](Cur& c) {
    if (++c.kt == KT) { c.kt = 0; c.it++; tile_base(c); }
  };
  auto stage_a = [&](const Cur& c, int par) {      // both A halves of the cursor's K-tile -> buffer `par`
    if (c.it >= nmine) return;
    char* dst = smem + par * 65536 + wave * 2048;
    const char* src = c.pa + c.kt * 128;
#pragma unroll
    for (int hh = 0; hh < 2; hh++)
#pragma unroll
      for (int i = 0; i < 2; i++) GLDS16(src + (size_t)hh * 128 * lda * 2 + voa[i], dst + hh * 16384 + i * 1024);
  };
  auto stage_w = [&](const Cur& c, int par) {
    if (c.it >= nmine) return;
    char* dst = smem + par * 65536 + 32768 + wave * 2048;
    const char* src = c.pw + c.kt * 128;
#pragma unroll
    for (int hh = 0; hh < 2; hh++)
#pragma unroll
      for (int i = 0; i < 2; i++) GLDS16(src + (size_t)hh * 128 * ldw * 2 + vow[i], dst + hh * 16384 + i * 1024);
  };
  f32x4 acc[2][4][2][2];   // [row half][16-row tile][column half][16-column tile]
#pragma unroll
  for (int a = 0; a < 2; a++)
#pragma unroll
    for (int b = 0; b < 4; b++)
#pragma unroll
      for (int c = 0; c < 2; c++)
#pragma unroll
        for (int d = 0; d < 2; d++) acc[a][b][c][d] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int fsw = (lane >> 1) & 7;
  const int aoff = wm * 16384 + (lane & 15) * 128;                               // + rh * 8192 + mi * 2048
  const int boff = 32768 + (wn >> 1) * 16384 + ((wn & 1) * 64 + (lane & 15)) * 128;   // + ch * 4096 + ni * 2048
  const int ch0 = ((lane >> 4) ^ fsw) << 4, ch1 = ((4 + (lane >> 4)) ^ fsw) << 4;     // the two k-steps' 16-byte chunks
  bf16x8 af[4][2], bfr[2][2][2];
  auto load_a = [&](const char* buf, int rh) {
#pragma unroll
    for (int mi = 0; mi < 4; mi++) {
      af[mi][0] = *reinterpret_cast<const bf16x8*>(buf + aoff + rh * 8192 + mi * 2048 + ch0);
      af[mi][1] = *reinterpret_cast<const bf16x8*>(buf + aoff + rh * 8192 + mi * 2048 + ch1);
    }
  };
  auto load_b = [&](const char* buf, int ch) {
#pragma unroll
    for (int ni = 0; ni < 2; ni++) {
      bfr[ch][ni][0] = *reinterpret_cast<const bf16x8*>(buf + boff + ch * 4096 + ni * 2048 + ch0);
      bfr[ch][ni][1] = *reinterpret_cast<const bf16x8*>(buf + boff + ch * 4096 + ni * 2048 + ch1);
    }
  };
#define SO3X_ROWS(RH)                                                                                                            \
  do {                                                                                                                           \
    __builtin_amdgcn_s_setprio(1);                                                                                               \
    _Pragma("unroll") for (int s_ = 0; s_ < 2; s_++) _Pragma("unroll") for (int c_ = 0; c_ < 2; c_++)                             \
    _Pragma("unroll") for (int mi = 0; mi < 4; mi++) _Pragma("unroll") for (int ni = 0; ni < 2; ni++)                             \
        acc[RH][mi][c_][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[c_][ni][s_], af[mi][s_], acc[RH][mi][c_][ni], 0, 0, 0);   \
    __builtin_amdgcn_s_setprio(0);                                                                                               \
  } while (0)
#define SO3X_H1_END()                                  \
  do {                                                 \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
    __builtin_amdgcn_s_barrier();                      \
  } while (0)
  // prologue: K-tile 0 whole, the W halves of K-tile 1
  Cur ca{0, 0, nullptr, nullptr}, cw{0, 0, nullptr, nullptr};
  tile_base(ca);
  tile_base(cw);
  stage_a(ca, 0); advance(ca);
  stage_w(cw, 0); advance(cw);
  stage_w(cw, 1); advance(cw);
  asm volatile("s_waitcnt vmcnt(4)" ::: "memory");     // (KT >= 2: the four pieces of W(1) are really there)
  __builtin_amdgcn_s_barrier();
  if (wm == 1) __builtin_amdgcn_s_barrier();
  int par = 0;
  bool after_store = false;   // the K-tile right behind a tile's stores: its A halves were staged ahead of them (below)
  for (int it = 0; it < nmine; it++) {
    for (int kt = 0; kt < KT; kt++, par ^= 1) {
      const char* buf = smem + par * 65536;
      load_a(buf, 0); load_b(buf, 0); load_b(buf, 1);
      if (!after_store) { stage_a(ca, par ^ 1); advance(ca); }
      SO3X_H1_END();
      SO3X_ROWS(0);
      __builtin_amdgcn_s_barrier();
      load_a(buf, 1);
      const bool more = cw.it < nmine;
      stage_w(cw, par); advance(cw);
      // vmcnt counts in issue order, stores included.  Behind a tile's 16 stores everything K-tile u+1 needs was issued BEFORE them
      // (its A halves ahead of the epilogue, below), so they may stay in flight with the W pieces staged just now: a wait that
      // had to retire them stalls every CU on the chip-wide burst of C at each tile boundary (measured: 25 % of the kernel).
      if (after_store) {
        if (more) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NST + 4) : "memory"); else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NST) : "memory");
        after_store = false;
      } else {
        if (more) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      SO3X_H1_END();
      SO3X_ROWS(1);
      __builtin_amdgcn_s_barrier();
    }
    // this tile's accumulators -> C (the next tile's first K-tiles are already on their way).  Through a wave-private 4 KB of LDS,
    // 16 rows at a time: the accumulators hold 4 consecutive columns of one row per lane (C^T = W A^T orientation) and go in as
    // one ds_write_b128 per 16 x 16 tile (the 16-byte unit XORed with the row: conflict-free both ways); they come back as 8
    // consecutive columns of a row per lane, so that one store instruction writes 8 rows x 128 contiguous bytes -- whole cache
    // lines.  (Storing from the accumulator layout directly -- 16 rows x 64 bytes per instruction at the row stride of C -- ran the
    // write side of a 8192^3 product at 0.5 TB/s: +35 % on the whole kernel.)
    const int tile = x_start + lb + it * nlb, tm = tile / ntn, tn = tile - tm * ntn;
    // the A halves of K-tile u+2, normally staged in PA.H1 of K-tile u+1, go out NOW, ahead of the stores (their buffer is the one
    // K-tile u just finished with: both groups are behind its last barrier)
    stage_a(ca, par ^ 1); advance(ca);
    after_store = true;
    char* ep = smem + 131072 + wave * 4096;
    const int g = lane >> 4, m_l = lane & 15, rrow = lane >> 3, c8 = lane & 7;
    const int ncol = tn * TB + wn * 64 + c8 * 8;
    const float4 bv0 = *reinterpret_cast<const float4*>(bias + ncol), bv1 = *reinterpret_cast<const float4*>(bias + ncol + 4);
    const size_t row0 = (size_t)tm * TB + wm * 128;
    // (the residual / mask rows are fetched FOUR units -- 8 rows x 128 bytes each -- ahead: vmcnt retires in issue order, so the
    //  wait for a load issued d units ago also retires every C store older than that; at d = 1 that was the store just issued
    //  (measured, same box: 152 -> 143 us on the 512 x 2048 residual product, 243 -> 215 us on the 2048 x 512 ReLU-mask one; all 16
    //  units up front costs the short-K products a bubble at every tile start))
    constexpr int SO3X_RQ = 4;
    bf16x8 rq[SO3X_RQ];   // unit u's rows sit in rq[u % SO3X_RQ]
    if constexpr (EPI == EPI_RESID || EPI == EPI_MASK) {
#pragma unroll
      for (int u = 0; u < SO3X_RQ; u++) rq[u] = *reinterpret_cast<const bf16x8*>(R + (row0 + 16 * (u >> 1) + rrow + 8 * (u & 1)) * ldr + ncol);
    }
    // LayerNorm folded in (GemmLN): the per-column vectors of this lane's eight columns, the rows' statistics one unit ahead
    float sv[8], gv[8], ev[8];
    if constexpr (LNM == LN_A) {
      const float4 a0 = *reinterpret_cast<const float4*>(ln.svec + ncol), a1 = *reinterpret_cast<const float4*>(ln.svec + ncol + 4);
      sv[0] = a0.x; sv[1] = a0.y; sv[2] = a0.z; sv[3] = a0.w; sv[4] = a1.x; sv[5] = a1.y; sv[6] = a1.z; sv[7] = a1.w;
    }
    if constexpr (LNM == LN_RESID) {
      const float4 a0 = *reinterpret_cast<const float4*>(ln.gamma + ncol), a1 = *reinterpret_cast<const float4*>(ln.gamma + ncol + 4);
      const float4 b0 = *reinterpret_cast<const float4*>(ln.beta + ncol), b1 = *reinterpret_cast<const float4*>(ln.beta + ncol + 4);
      gv[0] = a0.x; gv[1] = a0.y; gv[2] = a0.z; gv[3] = a0.w; gv[4] = a1.x; gv[5] = a1.y; gv[6] = a1.z; gv[7] = a1.w;
      ev[0] = b0.x; ev[1] = b0.y; ev[2] = b0.z; ev[3] = b0.w; ev[4] = b1.x; ev[5] = b1.y; ev[6] = b1.z; ev[7] = b1.w;
    }
    // the statistics of this wave's 128 rows: ONE load per tile (lane l: rows l and 64 + l), handed to the lanes that need a row by
    // ds_bpermute -- a load per 8-row unit sits behind that unit's C store in the vmcnt order, and waiting for it retires the
    // store (measured: +2 us per tile)
    float2 st_lo = {0.f, 1.f}, st_hi = {0.f, 1.f};
    if constexpr (LNM == LN_A || LNM == LN_RESID) {
      st_lo = *reinterpret_cast<const float2*>(ln.rowstat + 2 * (row0 + lane));
      st_hi = *reinterpret_cast<const float2*>(ln.rowstat + 2 * (row0 + 64 + lane));
    }
#pragma unroll
    for (int c = 0; c < 8; c++) {      // 16-row chunks: rh = c >> 2, mi = c & 3
      const int rh = c >> 2, mi = c & 3;
#pragma unroll
      for (int ch = 0; ch < 2; ch++)
#pragma unroll
        for (int ni = 0; ni < 2; ni++) {
          *reinterpret_cast<f32x4*>(ep + m_l * 256 + (((ch * 8 + ni * 4 + g) ^ m_l) << 4)) = acc[rh][mi][ch][ni];
          acc[rh][mi][ch][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int rr = 0; rr < 2; rr++) {
        const int row = rrow + 8 * rr;
        const f32x4 lo = *reinterpret_cast<const f32x4*>(ep + row * 256 + (((2 * c8) ^ row) << 4));
        const f32x4 hi = *reinterpret_cast<const f32x4*>(ep + row * 256 + (((2 * c8 + 1) ^ row) << 4));
        float v[8] = {lo[0] + bv0.x, lo[1] + bv0.y, lo[2] + bv0.z, lo[3] + bv0.w, hi[0] + bv1.x, hi[1] + bv1.y, hi[2] + bv1.z, hi[3] + bv1.w};
        const size_t grow = row0 + c * 16 + row;
        const int nu = 2 * c + rr + 1;                       // the next unit: rows 16 (nu >> 1) + rrow + 8 (nu & 1)
        float2 stc = {0.f, 1.f};
        if constexpr (LNM == LN_A || LNM == LN_RESID) {
          const int src = (c * 16 + row) & 63;               // row c * 16 + row of the wave's 128: lane src, low or high half by c
          stc.x = __shfl(c >= 4 ? st_hi.x : st_lo.x, src);
          stc.y = __shfl(c >= 4 ? st_hi.y : st_lo.y, src);
        }
        if constexpr (LNM == LN_A) {   // v = rstd (r W'^T) - rstd mean s + c   (bias = c)
          const float acc8[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          const float bc[8] = {bv0.x, bv0.y, bv0.z, bv0.w, bv1.x, bv1.y, bv1.z, bv1.w};
          const float nrm = -stc.y * stc.x;
#pragma unroll
          for (int e = 0; e < 8; e++) v[e] = fmaf(stc.y, acc8[e], fmaf(nrm, sv[e], bc[e]));
        }
        if constexpr (EPI == EPI_RELU) {
#pragma unroll
          for (int e = 0; e < 8; e++) v[e] = fmaxf(v[e], 0.f);
        }
        if constexpr (DROP && (EPI == EPI_RELU || EPI == EPI_RESID)) {
          {   // training-mode dropout: this lane's eight columns are one call's eight flags
            const uint32_t m = drop_keep8(gd.seed, ((uint64_t)grow * N + ncol) >> 3, gd.ctr_hi, gd.thr16);
#pragma unroll
            for (int e = 0; e < 8; e++) v[e] = (m >> e) & 1 ? v[e] * gd.scale : 0.f;
          }
        }
        if constexpr (EPI == EPI_RESID || EPI == EPI_MASK) {
          const int cu = 2 * c + rr, fu = cu + SO3X_RQ;     // this unit, the unit fetched now
          const bf16x8 rv = rq[cu % SO3X_RQ];
          if (fu < 16) rq[cu % SO3X_RQ] = *reinterpret_cast<const bf16x8*>(R + (row0 + 16 * (fu >> 1) + rrow + 8 * (fu & 1)) * ldr + ncol);
          if constexpr (LNM == LN_RESID) {   // the residual operand is the un-normalised row: its LayerNorm here
#pragma unroll
            for (int e = 0; e < 8; e++) v[e] += fmaf(((float)rv[e] - stc.x) * stc.y, gv[e], ev[e]);
          } else {
#pragma unroll
            for (int e = 0; e < 8; e++) v[e] = EPI == EPI_RESID ? v[e] + (float)rv[e] : ((float)rv[e] > 0.f ? (DROP ? v[e] * gd.scale : v[e]) : 0.f);
          }
        }
        if constexpr (LNM >= LN_STAT) {   // this row's sum and sum of squares over the wave's 64 columns (8 lanes x 8)
          float s1 = 0.f, s2 = 0.f;
#pragma unroll
          for (int e = 0; e < 8; e++) { s1 += v[e]; s2 = fmaf(v[e], v[e], s2); }
          s1 = sum8(s1);
          s2 = sum8(s2);
          if (c8 == 0) *reinterpret_cast<float2*>(ln.stat_part + ((grow * 8 + tn * 4 + wn) << 1)) = float2{s1, s2};
        }
        *reinterpret_cast<bf16x8*>(C + grow * ldc + ncol) =
            bf16x8{(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3], (bf16)v[4], (bf16)v[5], (bf16)v[6], (bf16)v[7]};
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
  if (wm == 0) __builtin_amdgcn_s_barrier();   // the barrier the other group started with
#undef SO3X_ROWS
#undef SO3X_H1_END
}

template <int EPI, bool DROP>
static int gemm_bf16_t(hipStream_t s, const bf16* A, int lda, const bf16* W, int ldw, bf16* C, int ldc, const float* bias, const bf16* R, int ldr,
                       int M, int N, int K, const GemmDrop& gd) {
  if (M % TB == 0 && N % TB == 0 && K >= 128 && (M / TB) * (N / TB) >= 384) {   // enough 256 x 256 tiles to keep 256 persistent workgroups busy
    const int ntiles = (M / TB) * (N / TB);
    hipLaunchKernelGGL((k_gemm256_bf16<EPI, DROP>), dim3((unsigned)(ntiles < 256 ? ntiles : 256)), dim3(512), 0, s, A, W, C, bias, R, M, N, K, lda, ldw, ldc, ldr, gd);
  } else {
    if ((M / BM) * (N / BN) < 1024) {   // fewer than two rounds of 128-row tiles (two workgroups per CU): 64-row tiles
      // (M % 128 == 0 is the entry point's contract, so 64 divides M)
      if (int rc = ensure_dyn_lds(g_gemm128[8 + 2 * EPI + (DROP ? 1 : 0)], (const void*)k_gemm_bf16<EPI, DROP, 2>, gemm128_lds<2>())) return rc;
      hipLaunchKernelGGL((k_gemm_bf16<EPI, DROP, 2>), dim3((unsigned)((M / 64) * (N / BN))), dim3(256), gemm128_lds<2>(), s, A, W, C, bias, R, M, N, K, lda, ldw, ldc, ldr, gd);
    } else {
      if (int rc = ensure_dyn_lds(g_gemm128[2 * EPI + (DROP ? 1 : 0)], (const void*)k_gemm_bf16<EPI, DROP, 4>, gemm128_lds<4>())) return rc;
      hipLaunchKernelGGL((k_gemm_bf16<EPI, DROP, 4>), dim3((unsigned)((M / BM) * (N / BN))), dim3(256), gemm128_lds<4>(), s, A, W, C, bias, R, M, N, K, lda, ldw, ldc, ldr, gd);
    }
  }
  return check_launch();
}
int gemm_bf16(hipStream_t s, const bf16* A, int lda, const bf16* W, int ldw, bf16* C, int ldc, const float* bias, const bf16* R, int ldr,
              int M, int N, int K, int epi, GemmDrop gd) {
  if (M % BM || N % BN || K % BK || !bias || ((epi == EPI_RESID || epi == EPI_MASK) && !R)) return SO3X_ERR_INVALID_ARG;
  // (the dropout forms are their own instantiations: the plain epilogues stay exactly the code they were)
  const bool drop = epi == EPI_MASK ? gd.scale != 1.f : gd.thr16 != 0;
#define SO3X_GEMM_CASE(E) \
  case E: return drop ? gemm_bf16_t<E, true>(s, A, lda, W, ldw, C, ldc, bias, R, ldr, M, N, K, gd) : gemm_bf16_t<E, false>(s, A, lda, W, ldw, C, ldc, bias, R, ldr, M, N, K, gd)
  switch (epi) {
    case EPI_NONE: return gemm_bf16_t<EPI_NONE, false>(s, A, lda, W, ldw, C, ldc, bias, R, ldr, M, N, K, gd);
    SO3X_GEMM_CASE(EPI_RELU);
    SO3X_GEMM_CASE(EPI_RESID);
    SO3X_GEMM_CASE(EPI_MASK);
  }
#undef SO3X_GEMM_CASE
  return SO3X_ERR_INVALID_ARG;
}

bool gemm_bf16_ln_ok(int M, int N, int K) { return M % TB == 0 && N % TB == 0 && K % BK == 0 && K >= 128 && (M / TB) * (N / TB) >= 384; }
// the LayerNorm-folded forms (GemmLN): the persistent kernel only; stat_part needs N == 512 (two column tiles x four waves)
int gemm_bf16_ln(hipStream_t s, const bf16* A, int lda, const bf16* W, int ldw, bf16* C, int ldc, const float* bias, const bf16* R, int ldr,
                 int M, int N, int K, int epi, int lnm, GemmLN ln) {
  if (!gemm_bf16_ln_ok(M, N, K) || !bias || (epi == EPI_RESID && !R)) return SO3X_ERR_INVALID_ARG;
  if ((lnm == LN_A || lnm == LN_RESID) && !ln.rowstat) return SO3X_ERR_INVALID_ARG;
  if (lnm >= LN_STAT && (N != 2 * TB || !ln.stat_part)) return SO3X_ERR_INVALID_ARG;
  const int ntiles = (M / TB) * (N / TB);
  const dim3 grid((unsigned)(ntiles < 256 ? ntiles : 256));
  const GemmDrop gd{};
#define SO3X_LN_LAUNCH(E, L) \
  hipLaunchKernelGGL((k_gemm256_bf16<E, false, L>), grid, dim3(512), 0, s, A, W, C, bias, R, M, N, K, lda, ldw, ldc, ldr, gd, ln)
  if (epi == EPI_NONE && lnm == LN_A) SO3X_LN_LAUNCH(EPI_NONE, LN_A);
  else if (epi == EPI_RELU && lnm == LN_A) SO3X_LN_LAUNCH(EPI_RELU, LN_A);
  else if (epi == EPI_RESID && lnm == LN_STAT) SO3X_LN_LAUNCH(EPI_RESID, LN_STAT);
  else if (epi == EPI_RESID && lnm == LN_RESID) SO3X_LN_LAUNCH(EPI_RESID, LN_RESID);
  else return SO3X_ERR_INVALID_ARG;
#undef SO3X_LN_LAUNCH
  return check_launch();
}

// ------------------------------------------------------------------------------------------------ attention forward
// DROP: training-mode dropout of the probabilities (torch: after the softmax).  The keep bits of this wave's 32 queries x the tile's
// 64 keys (maskq, so3x_planenet_bf16.hpp) arrive by one more LDS-DMA per wave and tile; the row sums (the softmax's denominator)
// take the plain probabilities, O^T the kept ones, and the 1 / keep goes into the final normalisation.
template <bool DROP>
__global__ __launch_bounds__(256, 2) void k_attn_fwd(const bf16* __restrict__ qkv, bf16* __restrict__ o, float* __restrict__ lse, int P,
                                                     float sc, float c2, const uint32_t* __restrict__ maskq, float inv_keep) {
  __shared__ __attribute__((aligned(16))) char smem[65536 + (DROP ? 2048 : 0)];   // 2 x (K tile 16 KB | V tile 16 KB) [| 2 x 4 waves x 64 mask words]
  typedef __attribute__((address_space(3))) s16x4* lds_p;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), r = lane & 31, h = lane >> 5;
  int xt, hd, b;
  attn_tile((P + 127) / 128, xt, hd, b);
  const int q0 = xt * 128 + wave * 32;
  const size_t tok0 = (size_t)b * P;
  // Q^T as the B operand of S^T = K Q^T: lane (r, h) holds Q[q0 + r][16 ks + 8 h .. + 8]
  bf16x8 qf[8];
  {
    const int qr = q0 + r < P ? q0 + r : P - 1;            // (P % 128 == 64: the last block's upper two waves have no queries)
    const bf16* qrow = qkv + (tok0 + qr) * (3 * D) + hd * DH + 8 * h;
#pragma unroll
    for (int ks = 0; ks < 8; ks++) qf[ks] = *reinterpret_cast<const bf16x8*>(qrow + 16 * ks);
  }
  const bf16* kbase = qkv + tok0 * (3 * D) + D + hd * DH;
  const bf16* vbase = kbase + D;
  const uint32_t* mrow = nullptr;   // lane (r, h): word 2 j + h of query r's row of keep bits
  if constexpr (DROP) mrow = maskq + (((size_t)b * HEADS + hd) * P + (q0 + r < P ? q0 + r : P - 1)) * (P / 32) + h;
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
  f32x16 ot[4];
#pragma unroll
  for (int dt = 0; dt < 4; dt++)
#pragma unroll
    for (int i = 0; i < 16; i++) ot[dt][i] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  constexpr float RESCALE_LOG2 = 4.0f;
  const bf16x8 ones = {(bf16)1.f, (bf16)1.f, (bf16)1.f, (bf16)1.f, (bf16)1.f, (bf16)1.f, (bf16)1.f, (bf16)1.f};
  // lane constants of the LDS reads
  const int krd = r * 256, ksw = swz16(r);                                   // K rows kb * 32 + r: swz16 does not see kb * 32
  const int g = lane >> 4, q_ = (lane & 15) >> 2, p_ = lane & 3;
  const int vrow0 = (4 * h + q_) * 256, vsub = (p_ & 1) * 8, vclo = 2 * (g & 1) + (p_ >> 1);
  const int nt = P / 64;
  stage(0, 0);
  // One tile; BUF is a compile-time constant (the loop below is unrolled by two): the compiler then sees that the tile it reads and
  // the tile the LDS-DMA of stage(j + 1) writes are disjoint ranges of smem, and does not put a `vmcnt(0)` -- a wait for the DMA
  // it has just issued -- in front of the V reads
  auto tile = [&](int j, auto buf_c) {
    constexpr int BUF = decltype(buf_c)::value;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (j + 1 < nt) stage(j + 1, BUF ^ 1);
    const char* sk = smem + BUF * 32768;
    const char* sv = sk + 16384;
    f32x16 st[2];
#pragma unroll
    for (int kb = 0; kb < 2; kb++) {
#pragma unroll
      for (int i = 0; i < 16; i++) st[kb][i] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 8; ks++) {
        const bf16x8 kf = *reinterpret_cast<const bf16x8*>(sk + kb * 8192 + krd + (((2 * ks + h) ^ ksw) << 4));
        st[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], st[kb], 0, 0, 0);
      }
    }
    // the K fragments six reads ahead of the MFMAs that take them (the scheduler's own order waits for every read in turn)
    __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
#pragma unroll
    for (int i = 0; i < 10; i++) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }
    __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
    // online softmax: this lane holds 32 of its query's 64 scores, lane ^ 32 the other 32.  The running maximum is raised -- and
    // O^T and the running sum rescaled -- only when some query of the wave outgrew it by more than 2^RESCALE_LOG2 (bf16 and fp32
    // are floating point: probabilities up to 2^4 lose nothing); the branch is wave-uniform.
    float mx = st[0][0];
#pragma unroll
    for (int i = 1; i < 16; i++) mx = fmaxf(mx, st[0][i]);
#pragma unroll
    for (int i = 0; i < 16; i++) mx = fmaxf(mx, st[1][i]);
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    if (!__all((mx - m_run) * c2 <= RESCALE_LOG2)) {
      const float m_new = fmaxf(m_run, mx);
      const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c2);
      m_run = m_new;
      l_run *= alpha;
#pragma unroll
      for (int dt = 0; dt < 4; dt++)
#pragma unroll
        for (int i = 0; i < 16; i++) ot[dt][i] *= alpha;
    }
    const float mneg = -m_run * c2;
#pragma unroll
    for (int kb = 0; kb < 2; kb++)
#pragma unroll
      for (int i = 0; i < 16; i++) st[kb][i] = __builtin_amdgcn_exp2f(fmaf(st[kb][i], c2, mneg));
    // O^T += V^T P^T: k-step s4 = keys 16 s4 .. 16 s4 + 15; the B operand is the accumulator registers 8 s' .. 8 s' + 7 as they sit
    // The row sums ride on the matrix pipe too: a fifth "V^T tile" of ones gives sum_k P[k][q] in every accumulator row -- four
    // MFMAs per tile instead of 64 vector adds (the vector port, not the matrix pipe, is this loop's bound), and the sum is of
    // the bf16 probabilities the numerator uses.
    f32x16 lt;
#pragma unroll
    for (int i = 0; i < 16; i++) lt[i] = 0.f;
    uint32_t wsh[2] = {0u, 0u};   // accumulator register i = key (i & 3) + 8 (i >> 2) + 4 h of its 32-key block: bit i' of word >> 4 h
    if constexpr (DROP) {
      const uint32_t* mw = reinterpret_cast<const uint32_t*>(smem + 65536 + BUF * 1024 + wave * 256);
      wsh[0] = mw[r] >> (4 * h);
      wsh[1] = mw[r + 32] >> (4 * h);
    }
#pragma unroll
    for (int s4 = 0; s4 < 4; s4++) {
      const int kb = s4 >> 1, s1 = s4 & 1;
      bf16x8 pf = {(bf16)st[kb][8 * s1 + 0], (bf16)st[kb][8 * s1 + 1], (bf16)st[kb][8 * s1 + 2], (bf16)st[kb][8 * s1 + 3],
                   (bf16)st[kb][8 * s1 + 4], (bf16)st[kb][8 * s1 + 5], (bf16)st[kb][8 * s1 + 6], (bf16)st[kb][8 * s1 + 7]};
      lt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones, pf, lt, 0, 0, 0);
      if constexpr (DROP) {
#pragma unroll
        for (int e = 0; e < 8; e++) {
          const int i = 8 * s1 + e;
          pf[e] = (wsh[kb] >> ((i & 3) + 8 * (i >> 2))) & 1u ? pf[e] : (bf16)0.f;
        }
      }
#pragma unroll
      for (int dt = 0; dt < 4; dt++) {
        // two 4-key x 16-column blocks per 16-lane group: keys 16 s4 + 4 h + (0..3) and + 8; columns 32 dt + 16 (g & 1) + (0..15)
        const int c0 = ((((dt ^ q_) << 2) | (vclo ^ h)) << 4) + vsub;
        const int c1 = ((((dt ^ q_) << 2) | (vclo ^ (h + 2))) << 4) + vsub;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(sv + s4 * 4096 + vrow0 + c0));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(sv + s4 * 4096 + vrow0 + 2048 + c1));
        const s16x8 v8 = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        ot[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, v8), pf, ot[dt], 0, 0, 0);
      }
    }
    l_run += lt[0];
  };
  for (int j = 0; j < nt; j += 2) {
    tile(j, std::integral_constant<int, 0>{});
    if (j + 1 < nt) tile(j + 1, std::integral_constant<int, 1>{});
  }
  const float l_tot = l_run;        // (the MFMA summed over all 64 keys of a tile: both lanes of a query hold the whole sum)
  const float inv = (DROP ? inv_keep : 1.f) / l_tot;
  // O^T (d in the registers, query on the lane) -> O rows through this wave's 8 KB of the (now idle) K / V buffers: 8-byte writes at
  // [query][d] with the 16-byte unit XORed with the query, 16-byte reads of whole rows, so that a store instruction writes 4 rows
  // x 256 contiguous bytes instead of 32 rows x 16
  __syncthreads();
  char* ob = smem + wave * 8192;
#pragma unroll
  for (int dt = 0; dt < 4; dt++)
#pragma unroll
    for (int g4 = 0; g4 < 4; g4++)
      *reinterpret_cast<bf16x4*>(ob + r * 256 + (((4 * dt + g4) ^ (r & 15)) << 4) + 8 * h) =
          bf16x4{(bf16)(ot[dt][4 * g4] * inv), (bf16)(ot[dt][4 * g4 + 1] * inv), (bf16)(ot[dt][4 * g4 + 2] * inv), (bf16)(ot[dt][4 * g4 + 3] * inv)};
  __builtin_amdgcn_wave_barrier();
  if (q0 >= P) return;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    const int qq = 4 * i + (lane >> 4), u = lane & 15;
    const bf16x8 v = *reinterpret_cast<const bf16x8*>(ob + qq * 256 + ((u ^ (qq & 15)) << 4));
    *reinterpret_cast<bf16x8*>(o + (tok0 + q0 + qq) * D + hd * DH + u * 8) = v;
  }
  if (lse && h == 0) lse[((size_t)b * HEADS + hd) * P + q0 + r] = m_run * sc + logf(l_tot);
}

// The attention dropout's keep bits of one layer in both layouts (LayerB): one wave per 64 queries x 64 keys, lane = query; element
// e = ((b H + h) P + q) P + k of the site (Drop), eight consecutive keys per Philox call; the transposed words by 64 ballots.
__global__ __launch_bounds__(256) void k_attn_mask(uint32_t* __restrict__ maskq, uint32_t* __restrict__ maskk, int P, uint32_t thr16, uint64_t seed,
                                                   uint64_t ctr_hi) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, qt = blockIdx.x, bh = blockIdx.y;
  const int q = qt * 64 + lane, wpr = P / 32;
  for (int kt = wave; kt < P / 64; kt += 4) {
    const uint64_t c0 = ((((uint64_t)bh * P + q) * P) + (uint64_t)kt * 64) >> 3;
    uint32_t lo = 0, hi = 0;
#pragma unroll
    for (int u = 0; u < 4; u++) {
      lo |= drop_keep8(seed, c0 + u, ctr_hi, thr16) << (8 * u);
      hi |= drop_keep8(seed, c0 + 4 + u, ctr_hi, thr16) << (8 * u);
    }
    uint32_t* rq = maskq + ((size_t)bh * P + q) * wpr + 2 * kt;
    rq[0] = lo;
    rq[1] = hi;
    unsigned long long mine = 0;
#pragma unroll 8
    for (int kk = 0; kk < 64; kk++) {
      const unsigned long long bal = __ballot((((kk < 32 ? lo : hi) >> (kk & 31)) & 1u) != 0u);
      if (lane == kk) mine = bal;
    }
    uint32_t* rk = maskk + ((size_t)bh * P + kt * 64 + lane) * wpr + 2 * qt;
    rk[0] = (uint32_t)mine;
    rk[1] = (uint32_t)(mine >> 32);
  }
}

// ------------------------------------------------------------------------------------------------ rows
// y = LayerNorm(r) gamma + beta over 512-wide bf16 rows; stats[n] = (mean, rstd); one wave per row, 16 bytes per lane
__global__ __launch_bounds__(256) void k_ln_bf16(const bf16* __restrict__ r, bf16* __restrict__ y, float* __restrict__ stats,
                                                 const float* __restrict__ gamma, const float* __restrict__ beta, int64_t rows, float eps) {
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  const bf16x8 v = *reinterpret_cast<const bf16x8*>(r + row * D + lane * 8);
  float x[8], s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; i++) { x[i] = (float)v[i]; s += x[i]; }
  const float mean = wave_sum(s) * (1.f / D);
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < 8; i++) { x[i] -= mean; q = fmaf(x[i], x[i], q); }
  const float rstd = 1.f / sqrtf(wave_sum(q) * (1.f / D) + eps);
  const float4 g0 = *reinterpret_cast<const float4*>(gamma + lane * 8), g1 = *reinterpret_cast<const float4*>(gamma + lane * 8 + 4);
  const float4 b0 = *reinterpret_cast<const float4*>(beta + lane * 8), b1 = *reinterpret_cast<const float4*>(beta + lane * 8 + 4);
  const bf16x8 out = {(bf16)(x[0] * rstd * g0.x + b0.x), (bf16)(x[1] * rstd * g0.y + b0.y), (bf16)(x[2] * rstd * g0.z + b0.z),
                      (bf16)(x[3] * rstd * g0.w + b0.w), (bf16)(x[4] * rstd * g1.x + b1.x), (bf16)(x[5] * rstd * g1.y + b1.y),
                      (bf16)(x[6] * rstd * g1.z + b1.z), (bf16)(x[7] * rstd * g1.w + b1.w)};
  *reinterpret_cast<bf16x8*>(y + row * D + lane * 8) = out;
  if (stats && lane == 0) {
    stats[row * 2] = mean;
    stats[row * 2 + 1] = rstd;
  }
}

// The time embedding of every cloud once (models.py:13-25; fp32 arithmetic as the exact form's k_embed): temb[b][j], j < 256
__global__ __launch_bounds__(256) void k_temb(const int64_t* __restrict__ t, float* __restrict__ temb, float neg_emb) {
  const int b = blockIdx.x, j = threadIdx.x, half = D2 / 2, jj = j < half ? j : j - half;
  const float f = (float)exp((double)((float)jj * neg_emb));
  const float arg = (float)t[b] * f;
  temb[b * D2 + j] = j < half ? sinf(arg) : cosf(arg);
}
// SIREN pre-activations (exact fp32, kept for the backward) and their sines (hardware sine: |pre| stays below ~200, where
// v_sin_f32 after the 1/2pi scaling is good to ~1e-5 -- two orders below the bf16 rounding of the result), the time embedding
// copied next to them; 8 columns per thread, rows >= N zero
__global__ __launch_bounds__(256) void k_embed_bf16(const float* __restrict__ x, const float* __restrict__ temb, const float* __restrict__ wp,
                                                    const float* __restrict__ bp, float* __restrict__ pre, bf16* __restrict__ sn,
                                                    bf16* __restrict__ h0, int64_t N, int64_t Npad, int64_t P) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= Npad * (D2 / 8)) return;
  const int64_t n = idx / (D2 / 8);
  const int j0 = (int)(idx - n * (D2 / 8)) * 8;
  bf16x8 sv, tv;
  if (n >= N) {
#pragma unroll
    for (int i = 0; i < 8; i++) { sv[i] = (bf16)0.f; tv[i] = (bf16)0.f; }
  } else {
    const float x0 = x[n * 3], x1 = x[n * 3 + 1], x2 = x[n * 3 + 2];
    const float* te = temb + (n / P) * D2 + j0;
    float a[8];
#pragma unroll
    for (int i = 0; i < 8; i++) {
      const int j = j0 + i;
      a[i] = fmaf(x2, wp[j * 3 + 2], fmaf(x1, wp[j * 3 + 1], fmaf(x0, wp[j * 3], bp[j])));
      const float rev = a[i] * 0.15915494309189535f;
      sv[i] = (bf16)__builtin_amdgcn_sinf(rev - floorf(rev));   // v_sin_f32 takes revolutions, |input| <= 256
      tv[i] = (bf16)te[i];
    }
    if (pre) {
      *reinterpret_cast<float4*>(pre + n * D2 + j0) = float4{a[0], a[1], a[2], a[3]};
      *reinterpret_cast<float4*>(pre + n * D2 + j0 + 4) = float4{a[4], a[5], a[6], a[7]};
    }
  }
  *reinterpret_cast<bf16x8*>(sn + n * D2 + j0) = sv;
  *reinterpret_cast<bf16x8*>(h0 + n * D + D2 + j0) = tv;
}

// PoolRN on bf16 rows (models.py:94-110): w_p = sigmoid(x_p . wpool + bpool);  xs_b = sum_p w_p x_p / max(sum_p w_p, 1e-6).
// One pass over the encoder output: grid (B clouds, PS point slices of 256), a wave per row (16 bytes per lane), the row's weight
// from a wave reduction, the weighted sum kept in 8 registers per lane; per workgroup part[b][slice][0..511 | 512 = sum of weights]
__global__ __launch_bounds__(256) void k_pool_bf16(const bf16* __restrict__ x, const float* __restrict__ wpool, const float* __restrict__ bpool,
                                                   float* __restrict__ w, float* __restrict__ part, int64_t P) {
  __shared__ float red[4][D + 1];
  const int b = blockIdx.x, sl = blockIdx.y, nsl = gridDim.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t p0 = (int64_t)sl * PSLICE, p1 = p0 + PSLICE < P ? p0 + PSLICE : P;
  const bf16* xb = x + (int64_t)b * P * D;
  float* wb = w + (int64_t)b * P;
  const float4 q0 = *reinterpret_cast<const float4*>(wpool + lane * 8), q1 = *reinterpret_cast<const float4*>(wpool + lane * 8 + 4);
  const float wp[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
  const float bias = bpool[0];
  float acc[8], sw = 0.f;
#pragma unroll
  for (int i = 0; i < 8; i++) acc[i] = 0.f;
  // four rows of a wave in flight (their loads, wave reductions and exponentials are independent chains; one row at a time the
  // loop is the sum of their latencies: 43 us for 64 MB), accumulated in row order
  for (int64_t p = p0 + wave; p < p1; p += 16) {
    bf16x8 v[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int64_t pu = p + 4 * u < p1 ? p + 4 * u : p;
      v[u] = *reinterpret_cast<const bf16x8*>(xb + pu * D + lane * 8);
    }
    float sdot[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < 8; i++) s = fmaf((float)v[u][i], wp[i], s);
      sdot[u] = s;
    }
#pragma unroll
    for (int u = 0; u < 4; u++) sdot[u] = wave_sum(sdot[u]) + bias;
#pragma unroll
    for (int u = 0; u < 4; u++) {
      if (p + 4 * u >= p1) break;
      const float wv = 1.f / (1.f + expf(-sdot[u]));
      if (lane == 0) wb[p + 4 * u] = wv;
      sw += wv;
#pragma unroll
      for (int i = 0; i < 8; i++) acc[i] = fmaf(wv, (float)v[u][i], acc[i]);
    }
  }
#pragma unroll
  for (int i = 0; i < 8; i++) red[wave][lane * 8 + i] = acc[i];
  if (lane == 0) red[wave][D] = sw;
  __syncthreads();
  float* out = part + ((int64_t)b * nsl + sl) * (D + 1);
  for (int c = threadIdx.x; c <= D; c += 256) out[c] = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
}
__global__ __launch_bounds__(512) void k_pool_final(const float* __restrict__ part, int nsl, float* __restrict__ S, float* __restrict__ xs) {
  const int b = blockIdx.x, c = threadIdx.x;
  const float* pb = part + (int64_t)b * nsl * (D + 1);
  float tot = 0.f, v = 0.f;
  for (int s = 0; s < nsl; s++) {
    tot += pb[(int64_t)s * (D + 1) + D];
    v += pb[(int64_t)s * (D + 1) + c];
  }
  xs[(int64_t)b * D + c] = v / fmaxf(tot, 1e-6f);
  if (c == 0) S[b] = tot;
}

size_t bf16_stash_bytes(const Shape& s) { return carve_b(s, nullptr, true).bytes; }

#define TRY(expr)                 \
  do {                            \
    int rc__ = (expr);            \
    if (rc__) return rc__;        \
  } while (0)
inline unsigned blocks_for(int64_t n, int per) { return (unsigned)((n + per - 1) / per); }

// W'[n][k] = bf16(W[n][k] gamma[k]), s[n] = sum_k W'[n][k] (of the ROUNDED values: what the matrix core multiplies), c[n] = sum_k
// W[n][k] beta[k] + bias[n]; 512-wide rows, one wave per row (GemmLN)
__global__ __launch_bounds__(256) void k_fold_ln(const float* __restrict__ W, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                 const float* __restrict__ bias, bf16* __restrict__ Wf, float* __restrict__ sv, float* __restrict__ cv,
                                                 int rows) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float* w = W + (size_t)row * D + lane * 8;
  float s1 = 0.f, c1 = 0.f;
  bf16x8 o;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    o[i] = (bf16)(w[i] * gamma[lane * 8 + i]);
    s1 += (float)o[i];
    c1 = fmaf(w[i], beta[lane * 8 + i], c1);
  }
  *reinterpret_cast<bf16x8*>(Wf + (size_t)row * D + lane * 8) = o;
  s1 = wave_sum(s1);
  c1 = wave_sum(c1);
  if (lane == 0) { sv[row] = s1; cv[row] = c1 + bias[row]; }
}
// (mean, rstd) of every row from the eight partial (sum, sum of squares) pairs a LN_STAT epilogue left
__global__ __launch_bounds__(256) void k_ln_stats_final(const float* __restrict__ part, float* __restrict__ stats, int64_t rows, float eps) {
  const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (row >= rows) return;
  const float4* p = reinterpret_cast<const float4*>(part + row * 16);
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int i = 0; i < 4; i++) { const float4 v = p[i]; s1 += v.x + v.z; s2 += v.y + v.w; }
  const float mean = s1 * (1.f / D), var = fmaxf(s2 * (1.f / D) - mean * mean, 0.f);
  *reinterpret_cast<float2*>(stats + row * 2) = float2{mean, 1.f / sqrtf(var + eps)};
}

size_t bf16_weights_bytes(const Shape& s) { return wimg_bytes(s); }

// the bf16 image of the weight matrices (everything up to the SIREN's post_scale weight; biases and LayerNorm vectors stay fp32)
// and the LayerNorm-folded copies of linear1 / in_proj with their vectors (wimg_bytes)
int weights_bf16(hipStream_t s, const Shape& sh, const float* prm, void* wimg, bool fold) {
  const ParamOff po = param_offsets(sh);
  const int64_t ncvt = (po.wps + (int64_t)D2 * D2 + 3) / 4;
  hipLaunchKernelGGL(k_cvt_bf16, dim3(blocks_for(ncvt, 256)), dim3(256), 0, s, prm, reinterpret_cast<bf16*>(wimg), ncvt);
  char* base = reinterpret_cast<char*>(wimg);
  for (int l = 0; fold && l < sh.L; l++) {   // (only the LayerNorm-folded inference path reads these: a training forward skips them)
    const LayerOff lo = po.layer(l);
    const FoldOff fo = fold_off(sh, l);
    hipLaunchKernelGGL(k_fold_ln, dim3(FF / 4), dim3(256), 0, s, prm + lo.w1, prm + lo.g1, prm + lo.be1, prm + lo.b1, reinterpret_cast<bf16*>(base + fo.w1),
                       reinterpret_cast<float*>(base + fo.s1), reinterpret_cast<float*>(base + fo.c1), FF);
    if (l > 0) {
      const LayerOff lp = po.layer(l - 1);
      hipLaunchKernelGGL(k_fold_ln, dim3(3 * D / 4), dim3(256), 0, s, prm + lo.wqkv, prm + lp.g2, prm + lp.be2, prm + lo.bqkv,
                         reinterpret_cast<bf16*>(base + fo.wqkv), reinterpret_cast<float*>(base + fo.sq), reinterpret_cast<float*>(base + fo.cq), 3 * D);
    }
  }
  return check_launch();
}

int forward_bf16(hipStream_t s, const Shape& sh, const float* prm, const float* x, const int64_t* t, float* out, float* encoding_out,
                 void* stash, void* workspace, const void* prepared, const Drop& dr) {
  const ParamOff po = param_offsets(sh);
  const int64_t N = sh.N(), Np = padded_rows(sh), P = sh.P;
  const bf16* wimg = prepared ? reinterpret_cast<const bf16*>(prepared) : reinterpret_cast<const bf16*>(workspace);
  const ActsB a = stash ? carve_b(sh, stash, true) : carve_b(sh, reinterpret_cast<char*>(workspace) + wimg_bytes(sh), false);
  // Inference at large token counts: the LayerNorms folded into the products around them (GemmLN) -- a.h[l + 1] then holds the
  // UN-normalised r2 of layer l with its statistics in st2, x1 is never formed, and only the last layer's output is normalised
  // (for the pooling).  Training (stash), dropout and small token counts take the plain sequence below.
  const bool folded = !stash && !dr.on() && gemm_bf16_ln_ok((int)Np, D, D) && gemm_bf16_ln_ok((int)Np, D, FF);
  if (!prepared) TRY(weights_bf16(s, sh, prm, workspace, folded));
  const float neg_emb = (float)(-(log(10000.0) / (D2 / 2 - 1)));
  hipLaunchKernelGGL(k_temb, dim3((unsigned)sh.B), dim3(D2), 0, s, t, a.temb, neg_emb);
  hipLaunchKernelGGL(k_embed_bf16, dim3(blocks_for(Np * (D2 / 8), 256)), dim3(256), 0, s, x, a.temb, prm + po.wp, prm + po.bp, a.pre, a.sn, a.h[0], N, Np, P);
  TRY(check_launch());
  TRY(gemm_bf16(s, a.sn, D2, wimg + po.wps, D2, a.h[0], D, prm + po.bps, nullptr, 0, (int)Np, D2, D2, EPI_NONE));   // post_scale -> h0[:, :256]
  const float sc = 1.f / sqrtf((float)DH), c2 = sc * 1.4426950408889634f;
  const char* wbase = reinterpret_cast<const char*>(wimg);
  for (int l = 0; folded && l < sh.L; l++) {
    const LayerOff lo = po.layer(l);
    const LayerB& k = a.layer[l];
    const bf16* h = a.h[l];           // l == 0: the embedding; else r2 of layer l - 1 (statistics: st2 of that layer)
    const FoldOff fo = fold_off(sh, l);
    const unsigned sblocks = blocks_for(Np, 256);
    GemmLN prev;                      // the LayerNorm between layer l - 1 and this one
    if (l > 0) {
      const LayerOff lp = po.layer(l - 1);
      prev.rowstat = a.layer[l - 1].st2;
      prev.svec = reinterpret_cast<const float*>(wbase + fo.sq);
      prev.gamma = prm + lp.g2;
      prev.beta = prm + lp.be2;
    }
    prev.stat_part = a.stat_part;
    if (l == 0) TRY(gemm_bf16(s, h, D, wimg + lo.wqkv, D, k.qkv, 3 * D, prm + lo.bqkv, nullptr, 0, (int)Np, 3 * D, D, EPI_NONE));
    else TRY(gemm_bf16_ln(s, h, D, reinterpret_cast<const bf16*>(wbase + fo.wqkv), D, k.qkv, 3 * D, reinterpret_cast<const float*>(wbase + fo.cq), nullptr, 0,
                          (int)Np, 3 * D, D, EPI_NONE, LN_A, prev));
    hipLaunchKernelGGL(k_attn_fwd<false>, dim3((unsigned)((P + 127) / 128 * HEADS * sh.B)), dim3(256), 0, s, k.qkv, k.o, k.lse, (int)P, sc, c2,
                       (const uint32_t*)nullptr, 1.f);
    TRY(check_launch());
    if (Np > N) {
      hipError_t e = hipMemsetAsync(k.o + N * D, 0, (size_t)(Np - N) * D * sizeof(bf16), s);
      if (e != hipSuccess) return (int)e;
    }
    TRY(gemm_bf16_ln(s, k.o, D, wimg + lo.wo, D, k.r1, D, prm + lo.bo, h, D, (int)Np, D, D, EPI_RESID, l == 0 ? LN_STAT : LN_RESID, prev));
    hipLaunchKernelGGL(k_ln_stats_final, dim3(sblocks), dim3(256), 0, s, a.stat_part, k.st1, Np, 1e-5f);
    GemmLN n1;                        // norm1 of this layer: folded into linear1, recomputed as linear2's residual
    n1.rowstat = k.st1;
    n1.svec = reinterpret_cast<const float*>(wbase + fo.s1);
    n1.gamma = prm + lo.g1;
    n1.beta = prm + lo.be1;
    n1.stat_part = a.stat_part;
    TRY(gemm_bf16_ln(s, k.r1, D, reinterpret_cast<const bf16*>(wbase + fo.w1), D, k.f, FF, reinterpret_cast<const float*>(wbase + fo.c1), nullptr, 0, (int)Np,
                     FF, D, EPI_RELU, LN_A, n1));
    TRY(gemm_bf16_ln(s, k.f, FF, wimg + lo.w2, FF, a.h[l + 1], D, prm + lo.b2, k.r1, D, (int)Np, D, FF, EPI_RESID, LN_RESID, n1));
    hipLaunchKernelGGL(k_ln_stats_final, dim3(sblocks), dim3(256), 0, s, a.stat_part, k.st2, Np, 1e-5f);
    TRY(check_launch());
    if (l == sh.L - 1) {              // the encoder's output itself: normalised in place
      hipLaunchKernelGGL(k_ln_bf16, dim3(blocks_for(Np, 4)), dim3(256), 0, s, a.h[l + 1], a.h[l + 1], (float*)nullptr, prm + lo.g2, prm + lo.be2, Np, 1e-5f);
      TRY(check_launch());
    }
  }
  for (int l = 0; !folded && l < sh.L; l++) {
    const LayerOff lo = po.layer(l);
    const LayerB& k = a.layer[l];
    const bf16* h = a.h[l];
    TRY(gemm_bf16(s, h, D, wimg + lo.wqkv, D, k.qkv, 3 * D, prm + lo.bqkv, nullptr, 0, (int)Np, 3 * D, D, EPI_NONE));
    const dim3 ag((unsigned)((P + 127) / 128 * HEADS * sh.B));
    if (dr.on()) {   // (a training forward: the stash is there)
      hipLaunchKernelGGL(k_attn_mask, dim3((unsigned)(P / 64), (unsigned)(sh.B * HEADS)), dim3(256), 0, s, k.maskq, k.maskk, (int)P, dr.thr16(), dr.seed,
                         dr.ctr_hi(l, DROP_ATTN));
      hipLaunchKernelGGL(k_attn_fwd<true>, ag, dim3(256), 0, s, k.qkv, k.o, k.lse, (int)P, sc, c2, k.maskq, dr.inv_keep());
    } else {
      hipLaunchKernelGGL(k_attn_fwd<false>, ag, dim3(256), 0, s, k.qkv, k.o, k.lse, (int)P, sc, c2, (const uint32_t*)nullptr, 1.f);
    }
    TRY(check_launch());
    if (Np > N) {   // the pad rows of the attention output feed the next GEMM: keep them finite (zero)
      hipError_t e = hipMemsetAsync(k.o + N * D, 0, (size_t)(Np - N) * D * sizeof(bf16), s);
      if (e != hipSuccess) return (int)e;
    }
    TRY(gemm_bf16(s, k.o, D, wimg + lo.wo, D, k.r1, D, prm + lo.bo, h, D, (int)Np, D, D, EPI_RESID, gemm_drop(dr, l, DROP_BLOCK1)));
    hipLaunchKernelGGL(k_ln_bf16, dim3(blocks_for(Np, 4)), dim3(256), 0, s, k.r1, k.x1, k.st1, prm + lo.g1, prm + lo.be1, Np, 1e-5f);
    TRY(gemm_bf16(s, k.x1, D, wimg + lo.w1, D, k.f, FF, prm + lo.b1, nullptr, 0, (int)Np, FF, D, EPI_RELU, gemm_drop(dr, l, DROP_FFN)));
    TRY(gemm_bf16(s, k.f, FF, wimg + lo.w2, FF, k.r2, D, prm + lo.b2, k.x1, D, (int)Np, D, FF, EPI_RESID, gemm_drop(dr, l, DROP_BLOCK2)));
    hipLaunchKernelGGL(k_ln_bf16, dim3(blocks_for(Np, 4)), dim3(256), 0, s, k.r2, a.h[l + 1], k.st2, prm + lo.g2, prm + lo.be2, Np, 1e-5f);
    TRY(check_launch());
  }
  const bf16* enc = a.h[sh.L];
  if (encoding_out) hipLaunchKernelGGL(k_cvt_f32, dim3(blocks_for(N * D / 4, 256)), dim3(256), 0, s, enc, encoding_out, N * D / 4);
  const int nsl = (int)((P + PSLICE - 1) / PSLICE);
  hipLaunchKernelGGL(k_pool_bf16, dim3((unsigned)sh.B, nsl), dim3(256), 0, s, enc, prm + po.wpool, prm + po.bpool, a.w, a.part, P);
  hipLaunchKernelGGL(k_pool_final, dim3((unsigned)sh.B), dim3(D), 0, s, a.part, nsl, a.S, a.xs);
  TRY(check_launch());
  TRY(head(s, a.xs, prm + po.wlin, prm + po.blin, prm + po.wout, prm + po.bout, a.pooled, out, sh.B, D));
  return SO3X_OK;
}

}  // namespace plane
}  // namespace so3x
