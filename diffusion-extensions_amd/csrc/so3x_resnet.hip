// so3x_resnet.hip -- the wide residual score network of so3_lock_train.py:11-59
//   RotPredict(d_model = 255): x0 = [R(9), sin(123), cos(123)];  x_{l+1} = x_l + silu(W_l x_l + b_l), l = 0..5
//   (models.py:28-34 ResLayer);  out = W_out x_6 + b_out (3 values for out_type = "skewvec", 6 for "rotmat"),
// on the CDNA4 matrix cores, standalone forward and fused into the chain-resident reverse sampler.
//
// Shape of the problem: 392,448 parameters = 768 KiB as bf16 -- five times the LDS -- and 781,830 flop per
// sample, 23x the 65-wide network.  So, unlike so3x_mlp.hpp (weights resident in LDS, activations flowing),
//   * the ACTIVATIONS are stationary: a wave owns 32 samples and keeps their 256-row residual stream in
//     128 fp32 VGPRs for the whole network (rows 0..254 features, row 255 the constant one that carries
//     every bias as a weight column; W row 255 is zero, so silu(0) = 0 keeps it at one);
//   * the WEIGHTS stream: one "chunk" = one 32-row output tile of one layer = the 16 (bf16) or 128 (fp32)
//     MFMA A-fragments of its K = 256 contraction, laid out fragment-major by a prep kernel; every
//     workgroup walks the 49 chunks of the network through a 3-slot LDS ring filled by LDS-DMA
//     (global_load_lds_dwordx4, two chunks in flight across the per-chunk barrier; the image is
//     L2-resident and shared by all workgroups);
//   * operand layout as in so3x_mlp.hpp: weights = A, activations = B, an accumulator tile has the row
//     map row(reg, h) = (reg & 3) + 8 (reg >> 2) + 4 h, and the image's K order is permuted so that the
//     residual registers, packed to bf16 pairs, ARE the next B operand (no lane movement, no LDS);
//   * bf16: the two waves of a SIMD run half a chunk out of phase -- wave A: MFMAs(c) then SiLU(c),
//     wave B: SiLU(c-1) then MFMAs(c) -- so the matrix pipe and the VALU/transcendental pipe overlap
//     across the pair (roofline: MFMA-bound, 128 MFMAs vs 128 SiLUs per lane per layer);
//   * bf16 SiLU scale fold as in so3x_mlp.hpp: the image carries -log2(e) W, the residual update is one
//     fma with the constant -1/log2(e).
// fp32 (parity path): v_mfma_f32_32x32x2_f32, one wave per SIMD (the operand copy needs a second
// 128-register set).
#include "so3x_common.hpp"
#include "so3x_math.hpp"
#include "so3x_igso3.hpp"
#include "so3x_reverse_step.hpp"

using namespace so3x;

namespace {

constexpr int DW = SO3X_RESNET_D;  // 255
constexpr int NBLK = 6;
constexpr int NFREQ = 123;         // models.py:18-24 with dim = 246
constexpr int LAYER_STRIDE = DW * DW + DW;
constexpr int NPARAMS = NBLK * LAYER_STRIDE + 3 * DW + 3;
static_assert(NPARAMS == SO3X_RESNET_PARAMS, "param count");
// The output layer is 3 or 6 rows (n_out, a run-time argument).  Head output o sits in row o (o < 4) or o + 4 of the output
// tile: rows 0..3, 8, 9 = accumulator registers 0..5 of the LOWER lane half (one lane holds all of a sample's outputs),
// and K slots 0..5 of that half when dout is the B operand of dX_6 = W_out^T dout.
constexpr int NOUT_MAX = 6;
__host__ __device__ constexpr int nparams(int nout) { return NBLK * LAYER_STRIDE + nout * (DW + 1); }
static_assert(nparams(6) == SO3X_RESNET_PARAMS_ROTMAT, "param count");
__host__ __device__ constexpr int head_of_row(int row) { return row < 4 ? row : ((row == 8 || row == 9) ? row - 4 : -1); }
__host__ __device__ constexpr int row_of_head(int o) { return o < 4 ? o : o + 4; }
constexpr int NCHUNK = NBLK * 8 + 1;  // (layer, output tile) in stream order, then the 3-row output layer
constexpr int RING = 3;
constexpr float kFoldS = -1.44269504088896341f;

struct Freqs { float f[NFREQ]; };

using f32x16 = float __attribute__((ext_vector_type(16)));
using bf16x8 = __bf16 __attribute__((ext_vector_type(8)));
using u32x4 = uint32_t __attribute__((ext_vector_type(4)));

template <int PREC> __host__ __device__ constexpr int chunk_bytes() { return PREC == SO3X_PREC_F32 ? 32768 : 16384; }
template <int PREC> __host__ __device__ constexpr int n_waves() { return PREC == SO3X_PREC_F32 ? 4 : 8; }
// The first NRES tiles of the stream stay in LDS for the whole launch (behind the ring) instead of being DMA'd on every
// pass: LDS-DMA issue is ~19 % of the bf16 kernel's time, 6 of 49 tiles is what fits beside the ring with
// room to spare (3 x 16 KiB + 6 x 16 KiB = 144 KiB).  Sampling kernels only (bf16, no stash).
template <int PREC, bool STASH> __host__ __device__ constexpr int n_resident() { return (PREC == SO3X_PREC_BF16 && !STASH) ? 6 : 0; }
template <int PREC, bool STASH> __host__ __device__ constexpr int lds_bytes() { return (3 + n_resident<PREC, STASH>()) * chunk_bytes<PREC>(); }
template <int PREC> __host__ __device__ constexpr size_t image_bytes() { return (size_t)NCHUNK * chunk_bytes<PREC>(); }
__host__ __device__ constexpr int row_of(int reg, int h) { return (reg & 3) + 8 * (reg >> 2) + 4 * h; }

// ---- prep: weight image --------------------------------------------------------------------
// value of (chunk, output row o, input feature f): weights, the bias in column 255, zeros in the padding
__device__ __forceinline__ float wvalue(const float* __restrict__ params, int chunk, int o, int f, int nout) {
  if (chunk < NBLK * 8) {
    const float* W = params + (size_t)(chunk >> 3) * LAYER_STRIDE;
    if (o >= DW) return 0.0f;
    return f < DW ? W[o * DW + f] : W[DW * DW + o];
  }
  const float* W = params + (size_t)NBLK * LAYER_STRIDE;
  o = o < 32 ? head_of_row(o) : -1;
  if (o < 0 || o >= nout) return 0.0f;
  return f < DW ? W[o * DW + f] : W[nout * DW + o];
}

// one 16-byte piece per thread: bf16 piece = (k-step ks, lane) -> 8 elements, element j of lane half h is
// feature 16 ks + 8 (j >> 2) + 4 h + (j & 3);  fp32 piece = (group g, lane) -> the lane's values of k-steps
// 4g .. 4g+3, k-step k of half h is feature 32 (k >> 4) + row(k & 15, h).
template <int PREC>
__global__ void __launch_bounds__(256) k_resnet_image(const float* __restrict__ params, void* __restrict__ img, int nout) {
  const int chunk = blockIdx.x;
  constexpr int PIECES = chunk_bytes<PREC>() / 16;
  for (int p = threadIdx.x; p < PIECES; p += blockDim.x) {
    const int lane = p & 63, i = lane & 31, h = lane >> 5, q = p >> 6;
    const int o = 32 * (chunk < NBLK * 8 ? (chunk & 7) : 0) + i;
    if constexpr (PREC == SO3X_PREC_BF16) {
      const float sc = chunk < NBLK * 8 ? kFoldS : 1.0f;
      bf16x8 v;
#pragma unroll
      for (int j = 0; j < 8; j++) v[j] = (__bf16)(sc * wvalue(params, chunk, o, 16 * q + 8 * (j >> 2) + 4 * h + (j & 3), nout));
      reinterpret_cast<bf16x8*>(img)[(size_t)chunk * PIECES + p] = v;
    } else {
      float4 v;
      float* e = reinterpret_cast<float*>(&v);
#pragma unroll
      for (int u = 0; u < 4; u++) { const int k = 4 * q + u; e[u] = wvalue(params, chunk, o, 32 * (k >> 4) + row_of(k & 15, h), nout); }
      reinterpret_cast<float4*>(img)[(size_t)chunk * PIECES + p] = v;
    }
  }
}

// ---- prep: per-timestep input rows  x0tab[t][256] = [0 x 9 (rotation slots), sin(t f), cos(t f), 1] ------
// models.py:22-24: the angle is formed in fp32 (int64 t promoted), then sin / cos of that fp32 angle.
__global__ void __launch_bounds__(256) k_resnet_x0tab(Freqs fr, int T, float* __restrict__ tab) {
  const int t = blockIdx.x, r = threadIdx.x;
  float v = 0.0f;
  if (r >= 9 && r < 9 + 2 * NFREQ) {
    const int e = r - 9;
    float sn, cs;
    sincos_cw((float)t * fr.f[e < NFREQ ? e : e - NFREQ], &sn, &cs);
    v = e < NFREQ ? sn : cs;
  } else if (r == 255) {
    v = 1.0f;
  }
  tab[(size_t)t * 256 + r] = v;
}

// ---- the streaming forward -----------------------------------------------------------------
template <int PREC> struct Operand;  // what the MFMAs read as B: a copy of the layer input
template <> struct Operand<SO3X_PREC_BF16> { uint32_t hi[64]; };
template <> struct Operand<SO3X_PREC_F32> { float x[128]; };

template <int PREC> __device__ __forceinline__ void refresh(Operand<PREC>& op, const float (&xf)[128]) {
  if constexpr (PREC == SO3X_PREC_BF16) {
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int i = 0; i < 64; i++) {
      bf16x2 p = {(__bf16)xf[2 * i], (__bf16)xf[2 * i + 1]};
      op.hi[i] = __builtin_bit_cast(uint32_t, p);
    }
  } else {
#pragma unroll
    for (int i = 0; i < 128; i++) op.x[i] = xf[i];
  }
}

// LDS-DMA of one chunk: the block's waves split its 1 KiB fragments; lane l lands at base + 16 l.
template <int PREC>
__device__ __forceinline__ void issue_chunk(const char* __restrict__ gimg, char* ring, int chunk, int slot, int wave, int lane) {
  constexpr int CB = chunk_bytes<PREC>();
  constexpr int PER_WAVE = CB / 1024 / n_waves<PREC>();
  asm volatile("" : "+v"(lane));  // the per-lane source address is formed at the issue, not kept (or spilled) across the pass
  const char* src = gimg + (size_t)chunk * CB + (size_t)(wave * PER_WAVE) * 1024 + lane * 16;
  char* dst = ring + slot * CB + (wave * PER_WAVE) * 1024;
#pragma unroll
  for (int i = 0; i < PER_WAVE; i++)
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + i * 1024),
                                     (__attribute__((address_space(3))) void*)(dst + i * 1024), 16, 0, 0);
}
template <int PREC> __host__ __device__ constexpr int dma_per_chunk() { return chunk_bytes<PREC>() / 1024 / n_waves<PREC>(); }

// wait until at most `keep` of this wave's LDS-DMAs are outstanding, then the workgroup barrier: afterwards the
// oldest chunk of the ring is complete for every wave and the slot freed one chunk ago may be refilled.
// Raw s_barrier on purpose: __syncthreads() would add a vmcnt(0) fence and drain the chunks still in flight.
template <int KEEP> __device__ __forceinline__ void ring_sync() {
  asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(KEEP) : "memory");
}

template <int PREC>
__device__ __forceinline__ f32x16 chunk_mfma(const char* slotp, const Operand<PREC>& op, int lane) {
  f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if constexpr (PREC == SO3X_PREC_BF16) {
    // The A fragments are read PF MFMAs (PF x 32 cycles > the LDS latency under 8 reading waves) ahead of their use, with
    // explicit reads and counted waits: hipcc's own pipeline of this loop waits lgkmcnt(0) before every PF-th MFMA, i.e.
    // for the read it issued one instruction earlier (+3.5 % on the chain kernel).
    constexpr int PF = 6;
    const uint32_t a0 = (uint32_t)(uintptr_t)slotp + lane * 16;  // LDS byte address (low 32 bits of the generic pointer)
    u32x4 a[16];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the counted waits below assume nothing else is outstanding
#define SO3X_RD(K) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(a[K]) : "v"(a0), "n"((K) * 1024))
#define SO3X_MM(K, LEFT)                                                                                             \
  {                                                                                                                  \
    asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(a[K]) : "n"(LEFT));                                                  \
    const u32x4 b = {op.hi[4 * (K)], op.hi[4 * (K) + 1], op.hi[4 * (K) + 2], op.hi[4 * (K) + 3]};                     \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[K]), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0); \
  }
    SO3X_RD(0); SO3X_RD(1); SO3X_RD(2); SO3X_RD(3); SO3X_RD(4); SO3X_RD(5);
    SO3X_MM(0, 5) SO3X_RD(6);
    SO3X_MM(1, 5) SO3X_RD(7);
    SO3X_MM(2, 5) SO3X_RD(8);
    SO3X_MM(3, 5) SO3X_RD(9);
    SO3X_MM(4, 5) SO3X_RD(10);
    SO3X_MM(5, 5) SO3X_RD(11);
    SO3X_MM(6, 5) SO3X_RD(12);
    SO3X_MM(7, 5) SO3X_RD(13);
    SO3X_MM(8, 5) SO3X_RD(14);
    SO3X_MM(9, 5) SO3X_RD(15);
    SO3X_MM(10, 5) SO3X_MM(11, 4) SO3X_MM(12, 3) SO3X_MM(13, 2) SO3X_MM(14, 1) SO3X_MM(15, 0)
#undef SO3X_RD
#undef SO3X_MM
    static_assert(PF == 6, "the schedule above is written out for PF = 6");
  } else {
    const float4* A = reinterpret_cast<const float4*>(slotp);
#pragma unroll
    for (int g = 0; g < 32; g++) {
      if ((g & 3) == 0) __builtin_amdgcn_sched_barrier(0);
      const float4 a = A[g * 64 + lane];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, op.x[4 * g], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, op.x[4 * g + 1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, op.x[4 * g + 2], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, op.x[4 * g + 3], acc, 0, 0, 0);
    }
  }
  return acc;
}

// residual update of one output tile: x += silu(z)
template <int PREC> __device__ __forceinline__ void residual(const f32x16& acc, float (&xf)[128], int TO /* constant after unrolling */) {
#pragma unroll
  for (int r = 0; r < 16; r++) {
    if constexpr (PREC == SO3X_PREC_BF16) {
      const float y = acc[r];  // = -log2(e) z
      xf[16 * TO + r] = fmaf(y * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(y)), 1.0f / kFoldS, xf[16 * TO + r]);
    } else {
      const float z = acc[r];
      xf[16 * TO + r] = fmaf(z, sigmoid_f32(z), xf[16 * TO + r]);  // fp32 path (parity gate G5)
    }
  }
}

// position in the chunk stream of a workgroup: ring slot of the next chunk to CONSUME
struct Stream { int slot; };

// first two chunks of a pass; call once before the first forward() of a workgroup
template <int PREC, int NRES = 0>
__device__ __forceinline__ void stream_begin(const char* __restrict__ gimg, char* ring, Stream& st, int wave, int lane) {
  st.slot = 0;
  issue_chunk<PREC>(gimg, ring, NRES, 0, wave, lane);
  issue_chunk<PREC>(gimg, ring, NRES + 1, 1, wave, lane);
  if constexpr (NRES > 0) {  // resident tiles 0 .. NRES-1 behind the ring
    const float4* src = reinterpret_cast<const float4*>(gimg);
    float4* dst = reinterpret_cast<float4*>(ring + RING * chunk_bytes<PREC>());
    for (int i = threadIdx.x; i < NRES * chunk_bytes<PREC>() / 16; i += blockDim.x) dst[i] = src[i];
    __syncthreads();
  }
}

// The whole network on this wave's 32 samples.  xf = the input rows on entry (destroyed); v = the three outputs of
// sample column (lane & 31), valid in the lower lane half.  Block-collective (49 barriers).  `again` = another
// forward() follows in this workgroup: its first two chunks are requested while this pass drains.
// STASH (training): the layer inputs X_l (the packed bf16 operand registers, l = 0..6) and the pre-activations
// Y_l = -log2(e) Z_l (l = 0..5) of this wave's 32 samples are dumped as they are, 32 B per lane per 32-row tile, for the
// backward kernels: layout [tile][part p][column n][half h][8 bf16] -- the lane's first 16 bytes in the tile's first KiB, its
// second 16 in the second, so that EACH store instruction of a wave writes one contiguous KiB.  (Rounds 1-3 kept a lane's 32
// bytes together, [tile][n][h][16 bf16]: every instruction then filled half of every cache line -- 4.1 TB/s of stores against
// 5.6 for whole KiBs, tools/ab/store_pattern.hip -- and the training forward is store-bound: 1.09 -> 0.95 ms at 2^19 samples.
// A deeper weight ring -- five chunks in flight, so that a store has 1.3 us instead of 0.5 to be acknowledged before the
// in-order vmcnt wait for a chunk stalls on it -- changed nothing: 0.94 ms.)
struct StashPtr { char* x; char* y; size_t layer_stride; };  // this wave's 16-KiB blocks of layer 0; +layer_stride per layer

template <bool NT = false> __device__ __forceinline__ void stash_tile(char* blk, int to, int lane, const uint32_t* p8) {
  asm volatile("" : "+v"(lane));  // per-lane store address formed at the store, not kept (or spilled) across the pass
  u32x4* d = reinterpret_cast<u32x4*>(blk + to * 2048 + (2 * (lane & 31) + (lane >> 5)) * 16);
  // NT = non-temporal.  Every dump is written once and read once, 10 GB of them per step against a 768 KB weight image that every
  // workgroup re-reads from L2: all dump stores, the dX chain's Y loads and the dW kernel's LDS-DMAs carry the hint (round 4, as a
  // replayed graph at 2^19 samples, same box: X dumps 2.32 -> 2.12 ms, + Y and dZ dumps 2.07, + the Y loads 2.06, + the DMAs
  // 2.03).  Round 2 had tried the stores alone, in the old lane-own-32-bytes layout, and seen the step get SLOWER (2.94 -> 3.2).
  if constexpr (NT) {
    __builtin_nontemporal_store(u32x4{p8[0], p8[1], p8[2], p8[3]}, d);
    __builtin_nontemporal_store(u32x4{p8[4], p8[5], p8[6], p8[7]}, d + 64);
  } else {
    d[0] = u32x4{p8[0], p8[1], p8[2], p8[3]};
    d[64] = u32x4{p8[4], p8[5], p8[6], p8[7]};
  }
}
// fp32 training path: the same dump with 16 fp32 per lane per tile (64 B per lane, 4 KiB per wave-tile, 32 KiB per block)
__device__ __forceinline__ void stash_tile_f32(char* blk, int to, int lane, const float* v16) {
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  f32x4* d = reinterpret_cast<f32x4*>(blk + to * 4096 + (2 * (lane & 31) + (lane >> 5)) * 64);
#pragma unroll
  for (int i = 0; i < 4; i++) d[i] = f32x4{v16[4 * i], v16[4 * i + 1], v16[4 * i + 2], v16[4 * i + 3]};
}
__device__ __forceinline__ void load_tile_f32(const char* blk, int to, int lane, float* v16) {
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  const f32x4* s4 = reinterpret_cast<const f32x4*>(blk + to * 4096 + (2 * (lane & 31) + (lane >> 5)) * 64);
#pragma unroll
  for (int i = 0; i < 4; i++) { const f32x4 v = s4[i]; v16[4 * i] = v.x; v16[4 * i + 1] = v.y; v16[4 * i + 2] = v.z; v16[4 * i + 3] = v.w; }
}
__device__ __forceinline__ uint32_t pack2(float a, float b) {
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  bf16x2 p = {(__bf16)a, (__bf16)b};
  return __builtin_bit_cast(uint32_t, p);
}

template <int PREC, bool STASH = false>
__device__ __forceinline__ void forward(const char* __restrict__ gimg, char* ring, Stream& st, float (&xf)[128], float (&v)[NOUT_MAX],
                                        bool again, int wave, int lane, StashPtr sp = StashPtr{nullptr, nullptr, 0}) {
  constexpr int CB = chunk_bytes<PREC>();
  constexpr int NRES = n_resident<PREC, STASH>();
  // vmcnt counts this wave's DMAs and stash stores together, in issue order.  A chunk's DMAs are issued two chunk
  // iterations before they are waited for, so the wait may leave outstanding everything issued since: the next chunk's
  // DMAs plus the stash stores of the two iterations in between (SY per stash_y, SX per stash_x; which iterations carry a
  // stash_x follows from the early / late schedule below).  Draining the stores at every chunk (vmcnt(0)) instead made
  // the 3.4 GB of dumps add to the compute time rather than hide under it.
  constexpr int DMA = dma_per_chunk<PREC>();
#if defined(SO3X_AB_BUILD) && defined(RESNET_AB_NO_Y)
  // timing build (tools/ab/wide_recompute_bound.sh): the training forward WITHOUT its Y dumps -- what a dX chain that recomputes
  // Y_l = W_l X_l from the X dumps would leave of this kernel
  constexpr int SY = 0, SX = !STASH ? 0 : 8 * (PREC == SO3X_PREC_BF16 ? 2 : 4);
#else
  constexpr int SY = !STASH ? 0 : (PREC == SO3X_PREC_BF16 ? 2 : 4), SX = 8 * SY;
#endif
  constexpr int DMA_OUT = STASH ? 0 : DMA;  // the two waits around the output layer stay conservative (once per pass)
  auto stash_x = [&](const Operand<PREC>& o, int l) {
    if constexpr (STASH && PREC == SO3X_PREC_BF16) {
#pragma unroll
      for (int to = 0; to < 8; to++) stash_tile<true>(sp.x + l * sp.layer_stride, to, lane, &o.hi[8 * to]);
    } else if constexpr (STASH) {
#pragma unroll
      for (int to = 0; to < 8; to++) stash_tile_f32(sp.x + l * sp.layer_stride, to, lane, &o.x[16 * to]);
    }
  };
  auto stash_y = [&](const f32x16& a, int l, int to) {
#if defined(SO3X_AB_BUILD) && defined(RESNET_AB_NO_Y)
    return;
#endif
    if constexpr (STASH && PREC == SO3X_PREC_BF16) {
      uint32_t p8[8];
#pragma unroll
      for (int i = 0; i < 8; i++) p8[i] = pack2(a[2 * i], a[2 * i + 1]);
      stash_tile<true>(sp.y + l * sp.layer_stride, to, lane, p8);
    } else if constexpr (STASH) {
      float z[16];
#pragma unroll
      for (int i = 0; i < 16; i++) z[i] = a[i];
      stash_tile_f32(sp.y + l * sp.layer_stride, to, lane, z);
    }
  };
  const bool late = PREC == SO3X_PREC_BF16 && (wave >> 2);  // the SIMD partner that runs its SiLU half a chunk later
  Operand<PREC> op;
  refresh<PREC>(op, xf);
  stash_x(op, 0);
  f32x16 acc;
  int slot = st.slot;
  // the counted waits below assume that this wave's only outstanding memory operations are the ring's DMAs
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll 1
  for (int l = 0; l < NBLK; l++) {
#pragma unroll
    for (int to = 0; to < 8; to++) {
      const int c = 8 * l + to;
      const bool res = to < NRES && l == 0;  // an LDS-resident tile: no DMA, no barrier, no ring slot
      if (!res) {
        // chunk c landed everywhere (chunk c+1 may still be in flight); slot of chunk c-1 is free
        if constexpr (!STASH) ring_sync<DMA>();
        else if (l == 0 && to < 2) ring_sync<DMA>();                            // right after the drain above: nothing to allow for
        else if (!late) { if (to < 2) ring_sync<DMA + 2 * SY + SX>(); else ring_sync<DMA + 2 * SY>(); }  // stash_x at to == 7
        else { if ((to == 1 || to == 2) && l > 0) ring_sync<DMA + 2 * SY + SX>(); else ring_sync<DMA + 2 * SY>(); }  // stash_x at to == 0
        const int nslot = slot == 0 ? 2 : slot - 1;  // (slot + 2) % 3
        if (c + 2 < NCHUNK) issue_chunk<PREC>(gimg, ring, c + 2, nslot, wave, lane);
        else if (again) issue_chunk<PREC>(gimg, ring, c + 2 - NCHUNK + NRES, nslot, wave, lane);
      }
      if (late) {
        if (to > 0) residual<PREC>(acc, xf, to - 1);
        else if (l > 0) { residual<PREC>(acc, xf, 7); refresh<PREC>(op, xf); stash_x(op, l); }
      }
      acc = chunk_mfma<PREC>(res ? ring + (RING + c) * CB : ring + slot * CB, op, lane);
      stash_y(acc, l, to);
      if (!late) {
        residual<PREC>(acc, xf, to);
        if (to == 7) { refresh<PREC>(op, xf); stash_x(op, l + 1); }
      }
      if (!res) slot = slot == 2 ? 0 : slot + 1;
    }
  }
  // output layer: chunk 48.  Outstanding DMAs here: chunk 48 itself and, with `again`, chunk 0 of the next pass.
  if (again) ring_sync<DMA_OUT>(); else ring_sync<0>();
  if (again) issue_chunk<PREC>(gimg, ring, NRES + 1, slot == 0 ? 2 : slot - 1, wave, lane);
  if (late) { residual<PREC>(acc, xf, 7); refresh<PREC>(op, xf); stash_x(op, NBLK); }
  acc = chunk_mfma<PREC>(ring + slot * CB, op, lane);
#pragma unroll
  for (int k = 0; k < NOUT_MAX; k++) v[k] = acc[k];  // head outputs 0..5 = regs 0..5 of the lower half (head_of_row)
  st.slot = slot == 2 ? 0 : slot + 1;
}

// input rows of a sample column: the x0tab row of its timestep with the rotation entries dropped into rows 0..8
// (lower half: rows 0-3 and 8 = regs 0-3 and 4;  upper half: rows 4-7 = regs 0-3)
__device__ __forceinline__ void fill_input(float (&xf)[128], const float (&R)[9], const float* __restrict__ x0row, int h) {
#pragma unroll
  for (int tq = 0; tq < 32; tq++) {
    const float4 e = *reinterpret_cast<const float4*>(x0row + 8 * tq + 4 * h);  // rows 32 tin + 8 q + 4 h + (0..3), tq = 4 tin + q
    xf[4 * tq] = e.x; xf[4 * tq + 1] = e.y; xf[4 * tq + 2] = e.z; xf[4 * tq + 3] = e.w;
  }
#pragma unroll
  for (int r = 0; r < 4; r++) xf[r] = h ? R[4 + r] : R[r];
  xf[4] = h ? xf[4] : R[8];
}
// the same from the five rotation entries this lane half feeds (r5 = rows 0-3 and 8 in the lower half, rows 4-7 in the
// upper): the chain kernel forms them straight from its quaternion state, so no 3x3 matrix is live across the row loads
__device__ __forceinline__ void rot5_from_quat(const Quat& q, int h, float (&r5)[5]) {
  float R[9];
  rmat_from_quat(q, R);
#pragma unroll
  for (int r = 0; r < 4; r++) r5[r] = h ? R[4 + r] : R[r];
  r5[4] = R[8];
}
__device__ __forceinline__ void fill_input5(float (&xf)[128], const float (&r5)[5], const float* __restrict__ x0row, int h) {
#pragma unroll
  for (int tq = 0; tq < 32; tq++) {
    const float4 e = *reinterpret_cast<const float4*>(x0row + 8 * tq + 4 * h);
    xf[4 * tq] = e.x; xf[4 * tq + 1] = e.y; xf[4 * tq + 2] = e.z; xf[4 * tq + 3] = e.w;
  }
#pragma unroll
  for (int r = 0; r < 4; r++) xf[r] = r5[r];
  xf[4] = h ? xf[4] : r5[4];
}

// ---- standalone forward ----------------------------------------------------------------------
template <int PREC, bool STASH = false>
__global__ void __launch_bounds__(64 * n_waves<PREC>(), 1)
k_resnet_fwd(const void* __restrict__ gimg, const float* __restrict__ x0tab, int T, const float* __restrict__ R,
             const int64_t* __restrict__ t, int64_t t_stride, float* __restrict__ out, int64_t n, int nout, char* stash_x = nullptr,
             char* stash_y = nullptr, size_t layer_stride = 0) {
  extern __shared__ __attribute__((aligned(16))) char ring[];
  constexpr int NW = n_waves<PREC>();
  const int lane = threadIdx.x & 63, col = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // provably wave-uniform: what derives from it lives in SGPRs
  const int64_t ngroups = (n + 32 * NW - 1) / (32 * NW);
  Stream st;
  stream_begin<PREC, n_resident<PREC, STASH>()>(reinterpret_cast<const char*>(gimg), ring, st, wave, lane);
  for (int64_t g = blockIdx.x; g < ngroups; g += gridDim.x) {
    int64_t idx = (g * NW + wave) * 32 + col;
    const bool live = idx < n;
    if (!live) idx = n - 1;
    float Rm[9];
    load_rot9(R, idx, Rm);
    int64_t tt = t[idx * t_stride];
    tt = tt < 0 ? 0 : (tt >= T ? T - 1 : tt);
    float xf[128], v[NOUT_MAX];
    fill_input(xf, Rm, x0tab + tt * 256, h);
    const size_t blk = (size_t)(g * NW + wave) * (PREC == SO3X_PREC_F32 ? 32768 : 16384);  // this wave's 32-sample block within a layer of the stash
    forward<PREC, STASH>(reinterpret_cast<const char*>(gimg), ring, st, xf, v, g + gridDim.x < ngroups, wave, lane,
                         StashPtr{stash_x + blk, stash_y + blk, layer_stride});
    int cole = col;  // the output address is formed after the network (see k_resnet_chain)
    asm volatile("" : "+v"(cole));
    const int64_t idxe = (g * NW + wave) * 32 + cole;
    if (out && idxe < n && h == 0) {
      out[idxe * nout] = v[0]; out[idxe * nout + 1] = v[1]; out[idxe * nout + 2] = v[2];
      if (nout == 6) { out[idxe * 6 + 3] = v[3]; out[idxe * 6 + 4] = v[4]; out[idxe * 6 + 5] = v[5]; }
    }
  }
}

// ---- chain-resident reverse sampler with this network (diffusion.py:315-337, so3_lock_test.py:24-31) ------
// Both lanes of a sample column carry the same unit-quaternion state and do the same per-step rotation math
// (keyed noise -> identical results); the 3 network outputs are mirrored from the lower half each step.
template <int PREC>
__global__ void __launch_bounds__(64 * n_waves<PREC>(), 1)
k_resnet_chain(const void* __restrict__ gimg, const float* __restrict__ x0tab, const float* __restrict__ sched, int T,
               const float* __restrict__ trap_p, const uint16_t* __restrict__ guide_p, const float* __restrict__ x_in, float* __restrict__ x_out, int t_start,
               int n_steps, const float* __restrict__ axes, const float* __restrict__ unif, uint64_t seed, uint64_t rng_offset,
               int64_t index_base, int64_t n, const int64_t* __restrict__ t_dev) {
  extern __shared__ __attribute__((aligned(16))) char ring[];
  if (t_dev) {  // the first timestep read on the device (the caller's `t` tensor: no host copy, no synchronisation), clamped into the tables
    const int64_t tv = t_dev[0];
    t_start = (int)(tv < n_steps - 1 ? n_steps - 1 : (tv > T - 1 ? T - 1 : tv));
  }
  constexpr int NW = n_waves<PREC>();
  const int lane = threadIdx.x & 63, col = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // provably wave-uniform: what derives from it lives in SGPRs
  const int64_t ngroups = (n + 32 * NW - 1) / (32 * NW);
  Stream st;
  stream_begin<PREC, n_resident<PREC, false>()>(reinterpret_cast<const char*>(gimg), ring, st, wave, lane);
  for (int64_t g = blockIdx.x; g < ngroups; g += gridDim.x) {
    const int64_t idx = (g * NW + wave) * 32 + col;
    const bool live = idx < n;
    const int64_t idc = live ? idx : n - 1;
    Quat q;
    {
      float Rm[9];
      load_rot9(x_in, idc, Rm);
      q = quat_from_rmat(Rm);
    }
    const bool last_group = g + gridDim.x >= ngroups;
#pragma unroll 1
    for (int s = 0; s < n_steps; s++) {
      const int t = t_start - s;
      float xf[128], vo[NOUT_MAX], v[3], r5[5];
      rot5_from_quat(q, h, r5);
      fill_input5(xf, r5, x0tab + (size_t)t * 256, h);
      forward<PREC>(reinterpret_cast<const char*>(gimg), ring, st, xf, vo, !(last_group && s == n_steps - 1), wave, lane);
#pragma unroll
      for (int j = 0; j < 3; j++) v[j] = __shfl(vo[j], col);
      // Everything the reverse step derives from the sample index and the timestep (row addresses of the schedule, the CDF
      // row and its guide, the first Philox round, the explicit-draw addresses) is formed HERE, after the network: made
      // opaque, the index and the timestep cannot be hoisted above the 49 chunks, where they sat in a dozen spilled
      // registers (round 1: 132 B of scratch per lane, 54x the algorithmic HBM writes).
      int tr = t, colr = col;
      asm volatile("" : "+s"(tr), "+v"(colr));
      const int64_t idxr = (g * NW + wave) * 32 + colr;
      const int64_t idcr = idxr < n ? idxr : n - 1;
      q = reverse_step<PREC == SO3X_PREC_BF16>(q, v, sched, T, tr, trap_p, guide_p, axes, unif, idcr, seed, rng_offset,
                                               (uint64_t)(index_base + idxr));  // bf16: hardware sine / cosine (so3x_math.hpp)
    }
    float Rm[9];
    rmat_from_quat(qnormalize(q), Rm);
    int cole = col;  // (the store address likewise: formed after the chain)
    asm volatile("" : "+v"(cole));
    const int64_t idxe = (g * NW + wave) * 32 + cole;
    if (idxe < n && h == 0) store_rot9(x_out, idxe, Rm);
  }
}

// =============================================================================================
// Backward (training): gradients wrt the 392,448 parameters for a given dL/dout; bf16 operands first, fp32 variants below.
//   so3x_resnet_bwd = [forward with stash] -> k_resnet_bwd (dX chain, writes dZ_l) -> k_resnet_dw (dW_l = dZ_l^T X_l,
//   split over sample ranges, fp32 partials) -> k_resnet_dw_reduce (fixed-order sum into the flat gradient).
// All three per-sample streams (X_l, Y_l, dZ_l) live in the workspace in the register-dump layout of StashPtr.
// =============================================================================================
constexpr int NTILE_T = NBLK * 8;  // transposed-weight tiles, stream order: layers 5..0, input-row tile 0..7

// transposed image: tile (layer l, input-row tile ti): A[m][k] = W_l[o(k)][32 ti + m], k in the operand order of the packed
// dZ registers (the same permutation as the forward image); true (unscaled) weights; the constant-one row gets no gradient.
// Tiles NTILE_T .. NTILE_T + 7 (one k-step each, at 16-KiB pitch like the rest): the output layer transposed,
// A[m][k] = W_out[k][32 ti + m] for k < n_out (K slot k of the lower half = head output k) -- dX_6 = W_out^T dout as eight
// MFMAs instead of 384 loads and FMAs per lane.
__global__ void __launch_bounds__(256) k_resnet_image_t(const float* __restrict__ params, void* __restrict__ img, int nout) {
  const int tile = blockIdx.x, ti = tile & 7;
  if (tile >= NTILE_T) {
    const float* Wo = params + (size_t)NBLK * LAYER_STRIDE;
    if (threadIdx.x < 64) {
      const int lane = threadIdx.x, m = lane & 31, h = lane >> 5, i = 32 * ti + m;
      bf16x8 v;
#pragma unroll
      for (int j = 0; j < 8; j++) v[j] = (__bf16)((h == 0 && j < nout && i < DW) ? Wo[j * DW + i] : 0.0f);
      reinterpret_cast<bf16x8*>(img)[(size_t)tile * 1024 + lane] = v;
    }
    return;
  }
  const int l = NBLK - 1 - (tile >> 3);
  const float* W = params + (size_t)l * LAYER_STRIDE;
  for (int p = threadIdx.x; p < 1024; p += blockDim.x) {
    const int lane = p & 63, m = lane & 31, h = lane >> 5, ks = p >> 6;
    const int i = 32 * ti + m;
    bf16x8 v;
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const int o = 16 * ks + 8 * (j >> 2) + 4 * h + (j & 3);
      v[j] = (__bf16)((o < DW && i < DW) ? W[o * DW + i] : 0.0f);
    }
    reinterpret_cast<bf16x8*>(img)[(size_t)tile * 1024 + p] = v;
  }
}

__device__ __forceinline__ void load_tile(const char* blk, int to, int lane, uint32_t* p8) {
  const u32x4* s = reinterpret_cast<const u32x4*>(blk + to * 2048 + (2 * (lane & 31) + (lane >> 5)) * 16);
  const u32x4 a = __builtin_nontemporal_load(s), b = __builtin_nontemporal_load(s + 64);  // read once (the comment at stash_tile)
  p8[0] = a.x; p8[1] = a.y; p8[2] = a.z; p8[3] = a.w; p8[4] = b.x; p8[5] = b.y; p8[6] = b.z; p8[7] = b.w;
}
__device__ __forceinline__ float bf_lo(uint32_t u) { return __builtin_bit_cast(float, u << 16); }
__device__ __forceinline__ float bf_hi(uint32_t u) { return __builtin_bit_cast(float, u & 0xffff0000u); }

// dX chain of one workgroup pass (8 waves x 32 samples):  dX_6 = W_out^T dout;  for l = 5..0:
//   dZ_l = dX_{l+1} * silu'(Z_l)  (stored for k_resnet_dw),  dX_l = dX_{l+1} + W_l^T dZ_l  (MFMA, C = the dX tile itself).
// The transposed weight tiles stream through the same 3-slot LDS ring as the forward's.
__global__ void __launch_bounds__(512, 1)
k_resnet_bwd(const void* __restrict__ gimg_t, const float* __restrict__ dout, const char* __restrict__ stash_y, char* __restrict__ stash_dz, size_t layer_stride, int64_t n, int nout) {
  extern __shared__ __attribute__((aligned(16))) char ring[];
  constexpr int PREC = SO3X_PREC_BF16, CB = chunk_bytes<PREC>();
  const int lane = threadIdx.x & 63, col = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // provably wave-uniform: what derives from it lives in SGPRs
  const int64_t ngroups = (n + 255) / 256;
  const char* gimg = reinterpret_cast<const char*>(gimg_t);
  int slot = 0;
  issue_chunk<PREC>(gimg, ring, 0, 0, wave, lane);
  issue_chunk<PREC>(gimg, ring, 1, 1, wave, lane);
  for (int64_t g = blockIdx.x; g < ngroups; g += gridDim.x) {
    const int64_t idx = (g * 8 + wave) * 32 + col;
    const bool live = idx < n;
    const size_t blk = (size_t)(g * 8 + wave) * 16384;
    const bool again_group = g + gridDim.x < ngroups;
    float dd[NOUT_MAX] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (live) {
#pragma unroll
      for (int k = 0; k < NOUT_MAX; k++) dd[k] = k < nout ? dout[idx * nout + k] : 0.0f;
    }
    const uint32_t dpk[3] = {h ? 0u : pack2(dd[0], dd[1]), h ? 0u : pack2(dd[2], dd[3]), h ? 0u : pack2(dd[4], dd[5])};
    f32x16 dx[8];
#if defined(SO3X_AB_BUILD) && defined(RESNET_AB_2X)
    f32x16 twice = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#endif
    {  // dX_6 = W_out^T dout on the matrix cores: K slots 0..5 of one k-step carry dout (lower lane half), A from the image tail
      const u32x4 bq = {dpk[0], dpk[1], dpk[2], 0u};
      const bf16x8 bop = __builtin_bit_cast(bf16x8, bq);
      const bf16x8* Ah = reinterpret_cast<const bf16x8*>(gimg + (size_t)NTILE_T * 16384);
      const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ti = 0; ti < 8; ti++) dx[ti] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah[(size_t)ti * 1024 + lane], bop, zero, 0, 0, 0);
    }
    {  // dZ of the output layer = dout in tile rows 0..3, 8, 9 (regs 0..5 of the lower half)
      uint32_t p8[8] = {dpk[0], dpk[1], dpk[2], 0, 0, 0, 0, 0};
      stash_tile(stash_dz + NBLK * layer_stride + blk, 0, lane, p8);
    }
#pragma unroll 1
    for (int l = NBLK - 1; l >= 0; l--) {
      uint32_t dzop[64];
      // The layer's eight Y tiles through a four-deep register queue: three tiles' loads are in flight while one is worked on.
      // (One at a time -- rounds 2-3 -- every tile paid a whole HBM round trip, 48 of them in a pass: the kernel ran at 3.8 TB/s.)
      constexpr int YQ = 4;
      uint32_t yq[YQ][8];
#pragma unroll
      for (int to = 0; to < YQ - 1; to++) load_tile(stash_y + l * layer_stride + blk, to, lane, yq[to]);
#pragma unroll
      for (int to = 0; to < 8; to++) {
        __builtin_amdgcn_sched_barrier(0);
        if (to + YQ - 1 < 8) load_tile(stash_y + l * layer_stride + blk, to + YQ - 1, lane, yq[(to + YQ - 1) % YQ]);
        const uint32_t (&y8)[8] = yq[to % YQ];
#pragma unroll
        for (int i = 0; i < 8; i++) {
          float g2[2];
#pragma unroll
          for (int e = 0; e < 2; e++) {
            const float y = e ? bf_hi(y8[i]) : bf_lo(y8[i]);                  // = -log2(e) z
            const float sg = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(y));
            const float z = y * (1.0f / kFoldS);
            g2[e] = dx[to][2 * i + e] * (sg * fmaf(z, 1.0f - sg, 1.0f));      // silu'(z) = s (1 + z (1 - s))
          }
          dzop[8 * to + i] = pack2(g2[0], g2[1]);
        }
        stash_tile<true>(stash_dz + l * layer_stride + blk, to, lane, &dzop[8 * to]);
      }
      // no drain here: vmcnt is in issue order, so the waits for this layer's Y loads have already retired every older DMA,
      // and the only operations still in flight are the last tile's two dZ stores -- the counted waits below absorb them
#pragma unroll
      for (int ti = 0; ti < 8; ti++) {
        const int c = 8 * (NBLK - 1 - l) + ti;
        ring_sync<dma_per_chunk<PREC>()>();
        {
          const int nslot = slot == 0 ? 2 : slot - 1;
          if (c + 2 < NTILE_T) issue_chunk<PREC>(gimg, ring, c + 2, nslot, wave, lane);
          else if (again_group) issue_chunk<PREC>(gimg, ring, c + 2 - NTILE_T, nslot, wave, lane);
        }
        const bf16x8* A = reinterpret_cast<const bf16x8*>(ring + slot * CB);
        f32x16 a = dx[ti];
#pragma unroll
        for (int gk = 0; gk < 4; gk++) {
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int k = 4 * gk; k < 4 * gk + 4; k++) {
            const u32x4 b = {dzop[4 * k], dzop[4 * k + 1], dzop[4 * k + 2], dzop[4 * k + 3]};
            a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[k * 64 + lane], __builtin_bit_cast(bf16x8, b), a, 0, 0, 0);
          }
        }
        dx[ti] = a;
#if defined(SO3X_AB_BUILD) && defined(RESNET_AB_2X)
        // timing build (tools/ab/wide_recompute_bound.sh): this tile's sixteen MFMAs a second time -- the matrix work a chain that
        // recomputes Y_l = W_l X_l instead of reading it would add (its second weight stream and its X operand registers not counted)
        {
          f32x16 a2 = twice;
#pragma unroll
          for (int k = 0; k < 16; k++) {
            const u32x4 b = {dzop[4 * k], dzop[4 * k + 1], dzop[4 * k + 2], dzop[4 * k + 3]};
            a2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[k * 64 + lane], __builtin_bit_cast(bf16x8, b), a2, 0, 0, 0);
          }
          twice = a2;
        }
#endif
        slot = slot == 2 ? 0 : slot + 1;
      }
    }
#if defined(SO3X_AB_BUILD) && defined(RESNET_AB_2X)
    if (twice[0] == 12345.678f) stash_dz[0] = 1;   // keeps the second product alive
#endif
  }
}

// dW_l = sum over samples dZ_l^T X_l: one workgroup = one (layer, sample range); wave w owns output rows 32w..32w+31
// x all 256 columns (8 accumulator tiles).  Per 32-sample block the two 16-KiB register dumps go to LDS by LDS-DMA, KiB by KiB
// (a tile's second KiB 128 bytes further on, so that the two lane groups of a transposed read sit in different banks: a tile
// takes DUMP_TILE bytes there), and are read back as MFMA operands with ds_read_b64_tr_b16 (K = the sample index): the quad
// (sample s, features 4 fq .. 4 fq+3) of a dump, lane index 2 s + (fq & 1), quad j = (fq >> 1) & 3 of the lane's eight, sits at
//   (fq >> 3) DUMP_TILE + (j >> 1) DUMP_PART1 + (2 s + (fq & 1)) 16 + (j & 1) 8.
constexpr int DUMP_NT = 2;  // cache-policy bits of the dumps' LDS-DMA: non-temporal (read once; the comment at stash_tile)
constexpr int DUMP_PART1 = 1024 + 128, DUMP_TILE = 2048 + 128, DUMP_LDS = 8 * DUMP_TILE;
struct DumpReadLane { int off[2]; };  // [part]: samples +0 / +4
__device__ __forceinline__ DumpReadLane dump_read_lane(int lane) {
  const int hh = lane >> 5, l32 = lane & 31, G = l32 >> 4, q = (l32 & 15) >> 2, pp = l32 & 3;
  // feature quad within the 32-feature tile: fq = 4 G + pp, so j = (2 G + (pp >> 1)) & 3: j >> 1 = G, j & 1 = pp >> 1
  DumpReadLane L;
#pragma unroll
  for (int part = 0; part < 2; part++) L.off[part] = (2 * (8 * hh + q + 4 * part) + (pp & 1)) * 16 + (pp >> 1) * 8 + G * DUMP_PART1;
  return L;
}
// operand = 8 bf16: samples 16 ks + 8 (lane >> 5) + 0..7 of feature 32 ft + (lane & 31)
__device__ __forceinline__ bf16x8 dump_frag(const char* img, const DumpReadLane& L, int ft, int ks) {
  typedef short s16x4 __attribute__((ext_vector_type(4)));
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  typedef __attribute__((address_space(3))) s16x4* lds_p;
  const int konst = ft * DUMP_TILE + ks * 512;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(img + konst + L.off[0]));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(img + konst + L.off[1]));
  s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, v);
}

// ---- fp32 training path (parity path: exact fp32 MFMA, fp32 dumps; one wave per SIMD in the chain kernel) ----------
__global__ void __launch_bounds__(256) k_resnet_image_t_f32(const float* __restrict__ params, void* __restrict__ img) {
  const int tile = blockIdx.x, l = NBLK - 1 - (tile >> 3), ti = tile & 7;
  const float* W = params + (size_t)l * LAYER_STRIDE;
  for (int p = threadIdx.x; p < 2048; p += blockDim.x) {  // piece = (group g of 4 k-steps, lane)
    const int lane = p & 63, m = lane & 31, h = lane >> 5, g = p >> 6;
    const int i = 32 * ti + m;
    float4 v;
    float* e = reinterpret_cast<float*>(&v);
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int k = 4 * g + u, o = 32 * (k >> 4) + row_of(k & 15, h);
      e[u] = (o < DW && i < DW) ? W[o * DW + i] : 0.0f;
    }
    reinterpret_cast<float4*>(img)[(size_t)tile * 2048 + p] = v;
  }
}

__global__ void __launch_bounds__(256, 1)
k_resnet_bwd_f32(const void* __restrict__ gimg_t, const float* __restrict__ params, const float* __restrict__ dout,
                 const char* __restrict__ stash_y, char* __restrict__ stash_dz, size_t layer_stride, int64_t n, int nout) {
  extern __shared__ __attribute__((aligned(16))) char ring[];
  constexpr int PREC = SO3X_PREC_F32, CB = chunk_bytes<PREC>();
  const int lane = threadIdx.x & 63, col = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // provably wave-uniform: what derives from it lives in SGPRs
  const int64_t ngroups = (n + 127) / 128;
  const char* gimg = reinterpret_cast<const char*>(gimg_t);
  const float* Wout = params + (size_t)NBLK * LAYER_STRIDE;
  int slot = 0;
  issue_chunk<PREC>(gimg, ring, 0, 0, wave, lane);
  issue_chunk<PREC>(gimg, ring, 1, 1, wave, lane);
  for (int64_t g = blockIdx.x; g < ngroups; g += gridDim.x) {
    const int64_t idx = (g * 4 + wave) * 32 + col;
    const bool live = idx < n;
    const size_t blk = (size_t)(g * 4 + wave) * 32768;
    const bool again_group = g + gridDim.x < ngroups;
    float dd[NOUT_MAX] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (live) {
#pragma unroll
      for (int k = 0; k < NOUT_MAX; k++) dd[k] = k < nout ? dout[idx * nout + k] : 0.0f;
    }
    f32x16 dx[8];
    const float* Wo = Wout;
    asm volatile("" : "+s"(Wo));
#pragma unroll
    for (int tq = 0; tq < 32; tq++) {
      const int f0 = 8 * tq + 4 * h;
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int f = f0 + r;
        const float w0 = f < DW ? Wo[f] : 0.f, w1 = f < DW ? Wo[DW + f] : 0.f, w2 = f < DW ? Wo[2 * DW + f] : 0.f;
        float a = w0 * dd[0] + w1 * dd[1] + w2 * dd[2];
        if (nout == 6) {
          const float w3 = f < DW ? Wo[3 * DW + f] : 0.f, w4 = f < DW ? Wo[4 * DW + f] : 0.f, w5 = f < DW ? Wo[5 * DW + f] : 0.f;
          a += w3 * dd[3] + w4 * dd[4] + w5 * dd[5];
        }
        dx[tq >> 2][4 * (tq & 3) + r] = a;
      }
    }
    {
      float z16[16] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      if (h == 0) {
#pragma unroll
        for (int k = 0; k < NOUT_MAX; k++) z16[k] = dd[k];  // tile rows 0..3, 8, 9
      }
      stash_tile_f32(stash_dz + NBLK * layer_stride + blk, 0, lane, z16);
    }
#pragma unroll 1
    for (int l = NBLK - 1; l >= 0; l--) {
      float dz[128];
#pragma unroll
      for (int to = 0; to < 8; to++) {
        float z16[16];
        __builtin_amdgcn_sched_barrier(0);
        load_tile_f32(stash_y + l * layer_stride + blk, to, lane, z16);
#pragma unroll
        for (int i = 0; i < 16; i++) {
          const float z = z16[i];
          const float sg = sigmoid_f32(z);
          dz[16 * to + i] = dx[to][i] * (sg * (1.0f + z * (1.0f - sg)));
        }
        stash_tile_f32(stash_dz + l * layer_stride + blk, to, lane, &dz[16 * to]);
      }
      // (no drain: see k_resnet_bwd)
#pragma unroll
      for (int ti = 0; ti < 8; ti++) {
        const int c = 8 * (NBLK - 1 - l) + ti;
        ring_sync<dma_per_chunk<PREC>()>();
        {
          const int nslot = slot == 0 ? 2 : slot - 1;
          if (c + 2 < NTILE_T) issue_chunk<PREC>(gimg, ring, c + 2, nslot, wave, lane);
          else if (again_group) issue_chunk<PREC>(gimg, ring, c + 2 - NTILE_T, nslot, wave, lane);
        }
        const float4* A = reinterpret_cast<const float4*>(ring + slot * CB);
        f32x16 a = dx[ti];
#pragma unroll
        for (int gk = 0; gk < 32; gk++) {
          if ((gk & 3) == 0) __builtin_amdgcn_sched_barrier(0);
          const float4 w = A[gk * 64 + lane];
          a = __builtin_amdgcn_mfma_f32_32x32x2f32(w.x, dz[4 * gk], a, 0, 0, 0);
          a = __builtin_amdgcn_mfma_f32_32x32x2f32(w.y, dz[4 * gk + 1], a, 0, 0, 0);
          a = __builtin_amdgcn_mfma_f32_32x32x2f32(w.z, dz[4 * gk + 2], a, 0, 0, 0);
          a = __builtin_amdgcn_mfma_f32_32x32x2f32(w.w, dz[4 * gk + 3], a, 0, 0, 0);
        }
        dx[ti] = a;
        slot = slot == 2 ? 0 : slot + 1;
      }
    }
  }
}

// fp32 dW: the two 32-KiB dumps of a 32-sample block in LDS (single buffer), operands by ds_read_b32: the value
// (sample s, feature 32 ft + m) of a dump sits at  ft 4096 + (2 s + ((m >> 2) & 1)) 64 + ((m & 3) + 4 (m >> 3)) 4.
__global__ void __launch_bounds__(512, 1)
k_resnet_dw_f32(const char* __restrict__ stash_x, const char* __restrict__ stash_dz, size_t layer_stride, int64_t nblk32,
                float* __restrict__ partial, int splits) {
  extern __shared__ __attribute__((aligned(16))) char lds[];  // [X | dZ], 2 x 32 KiB
  const int l = blockIdx.x / splits, split = blockIdx.x % splits;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, m = lane & 31, h = lane >> 5;
  const int64_t per = (nblk32 + splits - 1) / splits;
  const int64_t b0 = split * per, b1 = b0 + per < nblk32 ? b0 + per : nblk32;
  const char* xs = stash_x + l * layer_stride;
  const char* ds = stash_dz + l * layer_stride;
  const int lb = (2 * h + ((m >> 2) & 1)) * 64 + ((m & 3) + 4 * (m >> 3)) * 4;  // + 256 per k-step (2 samples)
  f32x16 acc[8];
#pragma unroll
  for (int tj = 0; tj < 8; tj++)
#pragma unroll
    for (int r = 0; r < 16; r++) acc[tj][r] = 0.0f;
  const bool head = l == NBLK;
  for (int64_t b = b0; b < b1; b++) {
    const uint4* sx = reinterpret_cast<const uint4*>(xs + (size_t)b * 32768);
    const uint4* sd = reinterpret_cast<const uint4*>(ds + (size_t)b * 32768);
    uint4* dxl = reinterpret_cast<uint4*>(lds);
    uint4* ddl = reinterpret_cast<uint4*>(lds + 32768);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; i++) { dxl[threadIdx.x + 512 * i] = sx[threadIdx.x + 512 * i]; ddl[threadIdx.x + 512 * i] = sd[threadIdx.x + 512 * i]; }
    __syncthreads();
    if (!head || wave == 0) {
#pragma unroll 4
      for (int ks = 0; ks < 16; ks++) {
        const float a = *reinterpret_cast<const float*>(lds + 32768 + wave * 4096 + ks * 256 + lb);
#pragma unroll
        for (int tj = 0; tj < 8; tj++) {
          const float bv = *reinterpret_cast<const float*>(lds + tj * 4096 + ks * 256 + lb);
          acc[tj] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv, acc[tj], 0, 0, 0);
        }
      }
    }
  }
  float* P = partial + ((size_t)l * splits + split) * 65536;
#pragma unroll
  for (int tj = 0; tj < 8; tj++)
#pragma unroll
    for (int r = 0; r < 16; r++) P[(32 * wave + row_of(r, h)) * 256 + 32 * tj + m] = acc[tj][r];
}

constexpr int DW_SPLITS = 36;  // fp32 path: (6 + 1) layers x 36 sample ranges = 252 workgroups
// bf16 path: the output layer's dZ dump is ONE tile (2 KiB of a block's 16: rows 0..3, 8, 9 of tile 0 are the head's outputs), so a
// head workgroup moves 18 KiB per block where a hidden layer's moves 32: 6 x 39 + 22 = 256 workgroups, 13.4 MB each at 2^19
// samples (7 x 36 = 252 moved 14.6 MB each, the head's 36 reading 14 KiB of nothing per block)
constexpr int DW_SPLITS_BODY = 39, DW_SPLITS_HEAD = 22, DW_GRID = NBLK * DW_SPLITS_BODY + DW_SPLITS_HEAD;

__global__ void __launch_bounds__(512, 1)
k_resnet_dw(const char* __restrict__ stash_x, const char* __restrict__ stash_dz, size_t layer_stride, int64_t nblk32,
            float* __restrict__ partial) {
  // The dumps go to LDS unchanged, which is what LDS-DMA does best: a 4-slot ring of [X | dZ] block pairs (4 x 32 KiB),
  // three blocks in flight per workgroup (96 KiB: the one-block-ahead register prefetch this replaces read at 3.7 TB/s,
  // 2/3 of the latency-bandwidth product), one raw barrier per block with counted vmcnt waits.
  extern __shared__ __attribute__((aligned(16))) char lds[];  // [slot 4][X | dZ], DUMP_LDS bytes each
  const bool head = blockIdx.x >= NBLK * DW_SPLITS_BODY;  // output layer: only tile-row 0 carries a gradient
  const int l = head ? NBLK : blockIdx.x / DW_SPLITS_BODY, split = head ? blockIdx.x - NBLK * DW_SPLITS_BODY : blockIdx.x % DW_SPLITS_BODY;
  const int nsplit = head ? DW_SPLITS_HEAD : DW_SPLITS_BODY;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wi = wave >> 1, wj = wave & 1;
  const int64_t per = (nblk32 + nsplit - 1) / nsplit;
  const int64_t b0 = split * per, b1 = b0 + per < nblk32 ? b0 + per : nblk32;
  const char* xs = stash_x + l * layer_stride;
  const char* ds = stash_dz + l * layer_stride;
  const DumpReadLane RL = dump_read_lane(lane);
  f32x16 acc[8];
#pragma unroll
  for (int tj = 0; tj < 8; tj++)
#pragma unroll
    for (int r = 0; r < 16; r++) acc[tj][r] = 0.0f;
  // 4 DMA pieces per wave and block: pieces 2 wave, 2 wave + 1 of each 16-KiB dump (the head's dZ: tile 0 only, wave 0's two)
  const bool dz_mine = !head || wave == 0;
  auto issue = [&](int64_t blk, int slot) {
    const char* gx = xs + (size_t)blk * 16384 + (size_t)(2 * wave) * 1024 + lane * 16;
    const char* gd = ds + (size_t)blk * 16384 + (size_t)(2 * wave) * 1024 + lane * 16;
    char* lx = lds + slot * (2 * DUMP_LDS) + wave * DUMP_TILE;   // this wave brings tile `wave` of both dumps: its two KiBs
#pragma unroll
    for (int i = 0; i < 2; i++) {
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gx + i * 1024),
                                       (__attribute__((address_space(3))) void*)(lx + i * DUMP_PART1), 16, 0, DUMP_NT);
      if (dz_mine)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gd + i * 1024),
                                         (__attribute__((address_space(3))) void*)(lx + DUMP_LDS + i * DUMP_PART1), 16, 0, DUMP_NT);
    }
  };
  for (int u = 0; u < 3; u++)
    if (b0 + u < b1) issue(b0 + u, u);
  int slot = 0;
  for (int64_t blk = b0; blk < b1; blk++) {
    // blocks younger than `blk` already requested: min(2, b1 - 1 - blk) -> that many x 4 (2: a head wave without dZ pieces) of
    // this wave's DMAs may stay in flight
    const int64_t younger = b1 - 1 - blk;
    if (dz_mine) { if (younger >= 2) ring_sync<8>(); else if (younger == 1) ring_sync<4>(); else ring_sync<0>(); }
    else { if (younger >= 2) ring_sync<4>(); else if (younger == 1) ring_sync<2>(); else ring_sync<0>(); }
    if (blk + 3 < b1) issue(blk + 3, (slot + 3) & 3);  // the slot of block blk - 1: every wave is past it
    const char* img = lds + slot * (2 * DUMP_LDS);
    // a wave's eight tiles are a 2 x 4 block of the layer's 8 x 8 -- output-row tiles 2 wi, 2 wi + 1 by input-column tiles
    // 4 wj .. 4 wj + 3: six operand fragments per k-step for its eight MFMAs (one row of eight tiles took nine: the transposed
    // reads, 4.5 x the block's bytes out of LDS, were what kept the loader from HBM's rate -- tools/ab/dma_streams.hip)
    // The transposed reads are inline assembly with counted lgkmcnt waits, not the builtin: the compiler knows that LDS-DMA wrote
    // this memory and put `s_waitcnt vmcnt(0)` in front of the first builtin read of every block -- draining the two younger block
    // pairs still in flight, i.e. ONE pair in flight instead of three (4.1 TB/s; the loader alone reaches 6.2, tools/ab/dma_streams.hip).
    // Both k-steps' 24 reads go out first, the first k-step's MFMAs start when its 12 have returned.
    if (!head || wi == 0) {
      typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
      const uint32_t xb = (uint32_t)(uintptr_t)img + 4 * wj * DUMP_TILE, db = (uint32_t)(uintptr_t)img + DUMP_LDS + 2 * wi * DUMP_TILE;
      const uint32_t x0 = xb + RL.off[0], x1 = xb + RL.off[1], d0 = db + RL.off[0], d1 = db + RL.off[1];
      u32x2 fb[2][4][2], fa[2][2][2];  // [k-step][fragment][part: samples +0 / +4]
#define SO3X_TR(DST, ADDR, OFF) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(DST) : "v"(ADDR), "n"(OFF))
#pragma unroll
      for (int ks = 0; ks < 2; ks++) {
#pragma unroll
        for (int c = 0; c < 4; c++) {
          SO3X_TR(fb[ks][c][0], x0, c * DUMP_TILE + ks * 512);
          SO3X_TR(fb[ks][c][1], x1, c * DUMP_TILE + ks * 512);
        }
#pragma unroll
        for (int r = 0; r < 2; r++) {
          SO3X_TR(fa[ks][r][0], d0, r * DUMP_TILE + ks * 512);
          SO3X_TR(fa[ks][r][1], d1, r * DUMP_TILE + ks * 512);
        }
      }
#undef SO3X_TR
#pragma unroll
      for (int ks = 0; ks < 2; ks++) {
        if (ks == 0)
          asm volatile("s_waitcnt lgkmcnt(12)" : "+v"(fb[0][0][0]), "+v"(fb[0][0][1]), "+v"(fb[0][1][0]), "+v"(fb[0][1][1]), "+v"(fb[0][2][0]), "+v"(fb[0][2][1]),
                       "+v"(fb[0][3][0]), "+v"(fb[0][3][1]), "+v"(fa[0][0][0]), "+v"(fa[0][0][1]), "+v"(fa[0][1][0]), "+v"(fa[0][1][1]));
        else
          asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fb[1][0][0]), "+v"(fb[1][0][1]), "+v"(fb[1][1][0]), "+v"(fb[1][1][1]), "+v"(fb[1][2][0]), "+v"(fb[1][2][1]),
                       "+v"(fb[1][3][0]), "+v"(fb[1][3][1]), "+v"(fa[1][0][0]), "+v"(fa[1][0][1]), "+v"(fa[1][1][0]), "+v"(fa[1][1][1]));
#pragma unroll
        for (int r = 0; r < 2; r++) {
          const u32x4 av = {fa[ks][r][0][0], fa[ks][r][0][1], fa[ks][r][1][0], fa[ks][r][1][1]};
#pragma unroll
          for (int c = 0; c < 4; c++) {
            const u32x4 bv = {fb[ks][c][0][0], fb[ks][c][0][1], fb[ks][c][1][0], fb[ks][c][1][1]};
            acc[4 * r + c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av), __builtin_bit_cast(bf16x8, bv), acc[4 * r + c], 0, 0, 0);
          }
        }
      }
    }
    slot = (slot + 1) & 3;
  }
  // partial[l][split][o][f], o = 32 (2 wi + r) + row(reg, h), f = 32 (4 wj + c) + (lane & 31)
  float* P = partial + ((size_t)l * DW_SPLITS_BODY + split) * 65536;
  const int n32 = lane & 31, h = lane >> 5;
#pragma unroll
  for (int t = 0; t < 8; t++)
#pragma unroll
    for (int r = 0; r < 16; r++) P[(32 * (2 * wi + (t >> 2)) + row_of(r, h)) * 256 + 32 * (4 * wj + (t & 3)) + n32] = acc[t][r];
}

// fixed-order sum of the split partials into the flat gradient (state_dict order); column 255 = the bias
// partial[(l * splits_body + split)][256][256]; the head's `splits_head` behind the hidden layers'
__global__ void __launch_bounds__(256) k_resnet_dw_reduce(const float* __restrict__ partial, float* __restrict__ dparams, int nout,
                                                          int splits_body, int splits_head) {
  const int l = blockIdx.y, o = blockIdx.x, f = threadIdx.x;
  const int rows = l < NBLK ? DW : nout;
  if (o >= rows || f >= 256) return;
  const int prow = l < NBLK ? o : row_of_head(o);  // the head's outputs sit in tile rows 0..3, 8, 9
  const float* P = partial + (size_t)l * splits_body * 65536 + prow * 256 + f;
  const int ns = l < NBLK ? splits_body : splits_head;
  float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;  // independent chains: the loads overlap (fixed order: deterministic)
  int k = 0;
  for (; k + 3 < ns; k += 4) {
    s0 += P[(size_t)k * 65536]; s1 += P[(size_t)(k + 1) * 65536]; s2 += P[(size_t)(k + 2) * 65536]; s3 += P[(size_t)(k + 3) * 65536];
  }
  if (k < ns) s0 += P[(size_t)k * 65536];
  if (k + 1 < ns) s1 += P[(size_t)(k + 1) * 65536];
  if (k + 2 < ns) s2 += P[(size_t)(k + 2) * 65536];
  const float s = (s0 + s1) + (s2 + s3);
  float* base = dparams + (size_t)l * LAYER_STRIDE;
  if (f < DW) base[o * DW + f] = s;
  else base[rows * DW + o] = s;
}

const Freqs& host_freqs() {
  static const Freqs fr = [] {  // initialised once, thread-safely
    Freqs f;
    so3x_posemb_freqs(NFREQ, f.f);
    return f;
  }();
  return fr;
}

size_t x0tab_offset(int precision) {
  return precision == SO3X_PREC_F32 ? image_bytes<SO3X_PREC_F32>() : image_bytes<SO3X_PREC_BF16>();
}
size_t ws_bytes(int precision, int T) { return x0tab_offset(precision) + (size_t)(T > 0 ? T : 0) * 256 * sizeof(float); }

template <int PREC> int prep(hipStream_t s, const float* params, int T, void* ws, int nout = 3) {
  hipLaunchKernelGGL((k_resnet_image<PREC>), dim3(NCHUNK), dim3(256), 0, s, params, ws, nout);
  float* tab = reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + image_bytes<PREC>());
  hipLaunchKernelGGL(k_resnet_x0tab, dim3(T), dim3(256), 0, s, host_freqs(), T, tab);
  return check_launch();
}

template <typename K> int grid_cap(PerDevice& st, K kernel, int threads, int lds, int* cap) {
  return resident_blocks(st, reinterpret_cast<const void*>(kernel), threads, lds, cap);
}

template <int PREC>
int launch_fwd(hipStream_t s, const void* ws, int T, const float* R, const int64_t* t, int64_t t_stride, float* out, int64_t n, int nout) {
  constexpr int LDS = lds_bytes<PREC, false>(), THREADS = 64 * n_waves<PREC>();
  static PerDevice caps;  // resident workgroups per device, queried once each
  int cap = 0;
  if (int rc = grid_cap(caps, &k_resnet_fwd<PREC>, THREADS, LDS, &cap)) return rc;
  const int64_t ngroups = (n + THREADS / 2 - 1) / (THREADS / 2);
  const float* tab = reinterpret_cast<const float*>(reinterpret_cast<const char*>(ws) + image_bytes<PREC>());
  hipLaunchKernelGGL((k_resnet_fwd<PREC>), dim3((int)(ngroups < cap ? ngroups : cap)), dim3(THREADS), LDS, s, ws, tab, T, R, t,
                     t_stride, out, n, nout);
  return check_launch();
}

template <int PREC>
int launch_chain(hipStream_t s, const void* ws, const float* sched, int T, const float* trap_p, const uint16_t* guide_p,
                 const float* x_in, float* x_out,
                 int t_start, int n_steps, const float* axes, const float* unif, uint64_t seed, uint64_t rng_offset,
                 int64_t index_base, int64_t n, const int64_t* t_dev = nullptr) {
  constexpr int LDS = lds_bytes<PREC, false>(), THREADS = 64 * n_waves<PREC>();
  static PerDevice caps;
  int cap = 0;
  if (int rc = grid_cap(caps, &k_resnet_chain<PREC>, THREADS, LDS, &cap)) return rc;
  const int64_t ngroups = (n + THREADS / 2 - 1) / (THREADS / 2);
  const float* tab = reinterpret_cast<const float*>(reinterpret_cast<const char*>(ws) + image_bytes<PREC>());
  hipLaunchKernelGGL((k_resnet_chain<PREC>), dim3((int)(ngroups < cap ? ngroups : cap)), dim3(THREADS), LDS, s, ws, tab, sched, T,
                     trap_p, guide_p, x_in, x_out, t_start, n_steps, axes, unif, seed, rng_offset, index_base, n, t_dev);
  return check_launch();
}

// training workspace: [forward image | x0tab | transposed image | X dumps 7 layers | Y dumps 6 | dZ dumps 7 | dW partials]
struct TrainLayout { size_t img_t, x, y, dz, partial, end, layer_stride; int64_t nblk32; };
TrainLayout train_layout(int64_t n, int T, int precision) {
  const bool f32 = precision == SO3X_PREC_F32;
  TrainLayout L;
  L.nblk32 = f32 ? ((n + 127) / 128) * 4 : ((n + 255) / 256) * 8;   // whole workgroup passes
  L.layer_stride = (size_t)L.nblk32 * (f32 ? 32768 : 16384);
  L.img_t = (ws_bytes(precision, T) + 255) & ~(size_t)255;
  L.x = L.img_t + (size_t)(NTILE_T + 8) * (f32 ? 32768 : 16384);  // + the transposed output layer (bf16 path)
  L.y = L.x + 7 * L.layer_stride;
  L.dz = L.y + 6 * L.layer_stride;
  L.partial = L.dz + 7 * L.layer_stride;
  L.end = L.partial + (size_t)(DW_GRID > 7 * DW_SPLITS ? DW_GRID : 7 * DW_SPLITS) * 65536 * sizeof(float);
  return L;
}

// the forward-with-dumps launch shared by so3x_resnet_fwd_stash and (when the caller brings no stash) so3x_resnet_bwd
template <int PREC>
int launch_fwd_stash(hipStream_t s, char* ws, int T, const float* params, const float* R, const int64_t* t, int64_t t_stride,
                     float* out, int64_t n, int nout, char* x_dump, char* y_dump, size_t layer_stride) {
  constexpr int LDS = RING * chunk_bytes<PREC>(), THREADS = 64 * n_waves<PREC>();
  int rc = prep<PREC>(s, params, T, ws, nout);
  if (rc) return rc;
  static PerDevice caps;
  int cap = 0;
  if ((rc = grid_cap(caps, &k_resnet_fwd<PREC, true>, THREADS, LDS, &cap))) return rc;
  const int64_t ngroups = (n + THREADS / 2 - 1) / (THREADS / 2);
  const float* tab = reinterpret_cast<const float*>(ws + image_bytes<PREC>());
  hipLaunchKernelGGL((k_resnet_fwd<PREC, true>), dim3((int)(ngroups < cap ? ngroups : cap)), dim3(THREADS), LDS, s, (const void*)ws, tab,
                     T, R, t, t_stride, out, n, nout, x_dump, y_dump, layer_stride);
  return check_launch();
}

}  // namespace

extern "C" {

size_t so3x_resnet_train_workspace_bytes(int64_t n, int precision, int t_table) {
  return train_layout(n > 0 ? n : 0, t_table, precision == SO3X_PREC_F32 ? SO3X_PREC_F32 : SO3X_PREC_BF16).end;
}

size_t so3x_resnet_stash_bytes(int64_t n, int precision) {
  return 13 * train_layout(n > 0 ? n : 0, 1, precision == SO3X_PREC_F32 ? SO3X_PREC_F32 : SO3X_PREC_BF16).layer_stride;
}

int so3x_resnet_fwd_stash(so3x_stream_t s_, const float* params, const float* R, const int64_t* t, int64_t t_stride, float* out,
                          void* stash, int64_t n, int n_out, int precision, int t_table, void* workspace, size_t workspace_bytes) {
  if (n < 0 || t_table <= 0 || (t_stride != 0 && t_stride != 1) || (n && (!params || !R || !t || !out || !stash)) ||
      (n_out != 3 && n_out != 6))
    return SO3X_ERR_INVALID_ARG;
  if (precision != SO3X_PREC_BF16 && precision != SO3X_PREC_F32) return SO3X_ERR_UNSUPPORTED;
  if (!workspace || workspace_bytes < ws_bytes(precision, t_table)) return SO3X_ERR_WORKSPACE;
  if (n == 0) return SO3X_OK;
  const TrainLayout L = train_layout(n, t_table, precision);
  char* st = reinterpret_cast<char*>(stash);
  char* ws = reinterpret_cast<char*>(workspace);
  return precision == SO3X_PREC_BF16
             ? launch_fwd_stash<SO3X_PREC_BF16>((hipStream_t)s_, ws, t_table, params, R, t, t_stride, out, n, n_out, st, st + 7 * L.layer_stride, L.layer_stride)
             : launch_fwd_stash<SO3X_PREC_F32>((hipStream_t)s_, ws, t_table, params, R, t, t_stride, out, n, n_out, st, st + 7 * L.layer_stride, L.layer_stride);
}

int so3x_resnet_bwd(so3x_stream_t s_, const float* params, const float* R, const int64_t* t, int64_t t_stride, const float* dout,
                    float* dparams, int64_t n, int n_out, int precision, int t_table, const void* stash, void* workspace,
                    size_t workspace_bytes) {
  if (n < 0 || t_table <= 0 || (t_stride != 0 && t_stride != 1) || !params || !dparams || (n && (!R || !t || !dout)) ||
      (n_out != 3 && n_out != 6))
    return SO3X_ERR_INVALID_ARG;
  if (precision != SO3X_PREC_BF16 && precision != SO3X_PREC_F32) return SO3X_ERR_UNSUPPORTED;
  const TrainLayout L = train_layout(n, t_table, precision);
  if (!workspace || workspace_bytes < L.end) return SO3X_ERR_WORKSPACE;
  hipStream_t s = (hipStream_t)s_;
  if (n == 0) return (int)hipMemsetAsync(dparams, 0, sizeof(float) * nparams(n_out), s);
  char* ws = reinterpret_cast<char*>(workspace);
  float* partial = reinterpret_cast<float*>(ws + L.partial);
  // layer inputs / pre-activations: the caller's stash (so3x_resnet_fwd_stash) or a forward run here into the workspace
  const char* xd = stash ? reinterpret_cast<const char*>(stash) : ws + L.x;
  const char* yd = stash ? xd + 7 * L.layer_stride : ws + L.y;
  int rc;
  if (precision == SO3X_PREC_BF16) {
    constexpr int PREC = SO3X_PREC_BF16, LDS = RING * chunk_bytes<PREC>();
    if (!stash && (rc = launch_fwd_stash<PREC>(s, ws, t_table, params, R, t, t_stride, nullptr, n, n_out, ws + L.x, ws + L.y, L.layer_stride))) return rc;
    hipLaunchKernelGGL(k_resnet_image_t, dim3(NTILE_T + 8), dim3(256), 0, s, params, (void*)(ws + L.img_t), n_out);
    static PerDevice caps_b, attr_dw;
    int cap_b = 0;
    if ((rc = grid_cap(caps_b, &k_resnet_bwd, 512, LDS, &cap_b))) return rc;
    const int64_t ngroups = (n + 255) / 256;
    hipLaunchKernelGGL(k_resnet_bwd, dim3((int)(ngroups < cap_b ? ngroups : cap_b)), dim3(512), LDS, s, (const void*)(ws + L.img_t),
                       dout, yd, ws + L.dz, L.layer_stride, n, n_out);
    if ((rc = ensure_dyn_lds(attr_dw, reinterpret_cast<const void*>(&k_resnet_dw), 8 * DUMP_LDS))) return rc;
    hipLaunchKernelGGL(k_resnet_dw, dim3(DW_GRID), dim3(512), 8 * DUMP_LDS, s, xd, (const char*)(ws + L.dz), L.layer_stride, L.nblk32,
                       partial);
  } else {
    constexpr int PREC = SO3X_PREC_F32, LDS = RING * chunk_bytes<PREC>();
    if (!stash && (rc = launch_fwd_stash<PREC>(s, ws, t_table, params, R, t, t_stride, nullptr, n, n_out, ws + L.x, ws + L.y, L.layer_stride))) return rc;
    hipLaunchKernelGGL(k_resnet_image_t_f32, dim3(NTILE_T), dim3(256), 0, s, params, (void*)(ws + L.img_t));
    static PerDevice caps_b, attr_dw;
    int cap_b = 0;
    if ((rc = grid_cap(caps_b, &k_resnet_bwd_f32, 256, LDS, &cap_b))) return rc;
    if ((rc = ensure_dyn_lds(attr_dw, reinterpret_cast<const void*>(&k_resnet_dw_f32), 65536))) return rc;
    const int64_t ngroups = (n + 127) / 128;
    hipLaunchKernelGGL(k_resnet_bwd_f32, dim3((int)(ngroups < cap_b ? ngroups : cap_b)), dim3(256), LDS, s, (const void*)(ws + L.img_t),
                       params, dout, yd, ws + L.dz, L.layer_stride, n, n_out);
    hipLaunchKernelGGL(k_resnet_dw_f32, dim3(7 * DW_SPLITS), dim3(512), 65536, s, xd, (const char*)(ws + L.dz), L.layer_stride, L.nblk32,
                       partial, DW_SPLITS);
  }
  const bool bf = precision == SO3X_PREC_BF16;
  hipLaunchKernelGGL(k_resnet_dw_reduce, dim3(256, 7), dim3(256), 0, s, (const float*)partial, dparams, n_out,
                     bf ? DW_SPLITS_BODY : DW_SPLITS, bf ? DW_SPLITS_HEAD : DW_SPLITS);
  return check_launch();
}

size_t so3x_resnet_workspace_bytes(int precision, int t_table) {
  return ws_bytes(precision == SO3X_PREC_F32 ? SO3X_PREC_F32 : SO3X_PREC_BF16, t_table);
}

int so3x_resnet_fwd(so3x_stream_t s, const float* params, const float* R, const int64_t* t, int64_t t_stride, float* out,
                    int64_t n, int n_out, int precision, int t_table, void* workspace, size_t workspace_bytes) {
  if (n < 0 || t_table <= 0 || (t_stride != 0 && t_stride != 1) || (n && (!params || !R || !t || !out)) ||
      (n_out != 3 && n_out != 6))
    return SO3X_ERR_INVALID_ARG;
  if (precision != SO3X_PREC_F32 && precision != SO3X_PREC_BF16) return SO3X_ERR_UNSUPPORTED;
  if (!workspace || workspace_bytes < ws_bytes(precision, t_table)) return SO3X_ERR_WORKSPACE;
  if (n == 0) return SO3X_OK;
  int rc = precision == SO3X_PREC_F32 ? prep<SO3X_PREC_F32>((hipStream_t)s, params, t_table, workspace, n_out)
                                      : prep<SO3X_PREC_BF16>((hipStream_t)s, params, t_table, workspace, n_out);
  if (rc) return rc;
  return precision == SO3X_PREC_F32 ? launch_fwd<SO3X_PREC_F32>((hipStream_t)s, workspace, t_table, R, t, t_stride, out, n, n_out)
                                    : launch_fwd<SO3X_PREC_BF16>((hipStream_t)s, workspace, t_table, R, t, t_stride, out, n, n_out);
}

int so3x_resnet_p_sample_chain(so3x_stream_t s, const float* params, const float* sched, int T, const float* trap_p,
                               const uint16_t* guide_p, const float* x_in, float* x_out, int t_start, int n_steps, const float* axes, const float* unif,
                               uint64_t seed, uint64_t rng_offset, int64_t index_base, int64_t n, int precision,
                               void* workspace, size_t workspace_bytes) {
  if (n < 0 || T <= 0 || n_steps < 0 || t_start < 0 || t_start >= T || t_start - n_steps + 1 < 0 ||
      (n && (!params || !sched || !trap_p || !x_in || !x_out)) || ((axes == nullptr) != (unif == nullptr)) ||
      (axes && n_steps > 1))
    return SO3X_ERR_INVALID_ARG;
  if (precision != SO3X_PREC_F32 && precision != SO3X_PREC_BF16) return SO3X_ERR_UNSUPPORTED;
  if (!workspace || workspace_bytes < ws_bytes(precision, T)) return SO3X_ERR_WORKSPACE;
  if (n == 0 || n_steps == 0) return SO3X_OK;
  int rc = precision == SO3X_PREC_F32 ? prep<SO3X_PREC_F32>((hipStream_t)s, params, T, workspace)
                                      : prep<SO3X_PREC_BF16>((hipStream_t)s, params, T, workspace);
  if (rc) return rc;
  if (precision == SO3X_PREC_F32)
    return launch_chain<SO3X_PREC_F32>((hipStream_t)s, workspace, sched, T, trap_p, guide_p, x_in, x_out, t_start, n_steps, axes, unif,
                                       seed, rng_offset, index_base, n);
  return launch_chain<SO3X_PREC_BF16>((hipStream_t)s, workspace, sched, T, trap_p, guide_p, x_in, x_out, t_start, n_steps, axes, unif,
                                      seed, rng_offset, index_base, n);
}

// The same in two calls for callers that drive the chain one step per call (so3_lock_test.py:24-31), as so3x_p_sample_prepare /
// so3x_p_sample_prepared do for the 65-wide network: the weight image and the [T][256] input-row table depend on the parameters only.
int so3x_resnet_p_sample_prepare(so3x_stream_t s, const float* params, int T, int precision, void* workspace, size_t workspace_bytes) {
  if (T <= 0 || !params) return SO3X_ERR_INVALID_ARG;
  if (precision != SO3X_PREC_F32 && precision != SO3X_PREC_BF16) return SO3X_ERR_UNSUPPORTED;
  if (!workspace || workspace_bytes < ws_bytes(precision, T)) return SO3X_ERR_WORKSPACE;
  return precision == SO3X_PREC_F32 ? prep<SO3X_PREC_F32>((hipStream_t)s, params, T, workspace) : prep<SO3X_PREC_BF16>((hipStream_t)s, params, T, workspace);
}

int so3x_resnet_p_sample_prepared(so3x_stream_t s, const float* sched, int T, const float* trap_p, const uint16_t* guide_p, const float* x_in,
                                  float* x_out, int t_start, const int64_t* t_dev, int n_steps, const float* axes, const float* unif, uint64_t seed,
                                  uint64_t rng_offset, int64_t index_base, int64_t n, int precision, const void* workspace,
                                  size_t workspace_bytes) {
  if (n < 0 || T <= 0 || n_steps < 0 || (n && (!sched || !trap_p || !x_in || !x_out)) || ((axes == nullptr) != (unif == nullptr)) ||
      (axes && n_steps > 1))
    return SO3X_ERR_INVALID_ARG;
  if (!t_dev && (t_start < 0 || t_start >= T || t_start - n_steps + 1 < 0)) return SO3X_ERR_INVALID_ARG;
  if (t_dev && n_steps > T) return SO3X_ERR_INVALID_ARG;
  if (precision != SO3X_PREC_F32 && precision != SO3X_PREC_BF16) return SO3X_ERR_UNSUPPORTED;
  if (!workspace || workspace_bytes < ws_bytes(precision, T)) return SO3X_ERR_WORKSPACE;
  if (n == 0 || n_steps == 0) return SO3X_OK;
  if (precision == SO3X_PREC_F32)
    return launch_chain<SO3X_PREC_F32>((hipStream_t)s, workspace, sched, T, trap_p, guide_p, x_in, x_out, t_start, n_steps, axes, unif, seed,
                                       rng_offset, index_base, n, t_dev);
  return launch_chain<SO3X_PREC_BF16>((hipStream_t)s, workspace, sched, T, trap_p, guide_p, x_in, x_out, t_start, n_steps, axes, unif, seed,
                                      rng_offset, index_base, n, t_dev);
}

}  // extern "C"
