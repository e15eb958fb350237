// so3x_mlp_bwd.hip -- backward of the RotPredict score network (autograd of
// so3_train.py:39-49; only the 17,358 parameters need gradients).
//
// Two paths.
// (A) bf16 operands + bounded timesteps (what SO3Diffusion.p_losses runs): ONE kernel, k_bwd_fused -- chain waves and dW
//     waves of a workgroup hand dZ_l / H_l over through LDS images, nothing per-sample returns to HBM; the forward's
//     pre-activations come from the training forward's stash (k_mlp_fwd_stash, 544 B/sample) or are recomputed.
//     Described at the kernel and in DESIGN.md section 4.
// (B) every other case (fp32; unbounded timesteps): three kernels per chunk of <= 2^19 samples (activations are
//     recomputed, never saved by the forward op):
//   K1 k_bwd_stage : per 32-sample tile, recompute the forward keeping the pre-activations
//                    Z_l in registers, then run the dZ chain  dH_{l-1} = W_l^T dZ_l  on the
//                    matrix cores (transposed-weight fragments, same accumulator-as-operand
//                    trick as the forward) and write dZ_l and the layer inputs H_l
//                    feature-major ([row][sample]) to the stash;
//   K2 k_bwd_dw    : dW_l = dZ_l . H_l^T, a skinny GEMM with K = samples: every block takes a
//                    sample range, transposes 32-sample slices through LDS into MFMA
//                    fragments, and its 8 waves own the 39 output tiles; partial sums go to
//                    one slab per block (deterministic, no float atomics);
//   K3 k_bwd_reduce: sums the slabs into dparams.
// The stash is fp32 / bf16 with the operand precision; K2 runs exact-fp32 or bf16 MFMA accordingly.
#include <stdlib.h>
#include <string.h>
#include "so3x_common.hpp"
#include "so3x_igso3.hpp"
#include "so3x_mlp.hpp"
#include "so3x_reverse_step.hpp"
#include "so3x_train.hpp"

using namespace so3x;
using namespace so3x::mlp;
using namespace so3x::train;

namespace {

constexpr int HROWS = 66;                       // 65 inputs + the constant one (bias carrier)
constexpr int H_BASE(int l) { return HROWS * l; }
constexpr int DZ_BASE(int l) { return 5 * HROWS + D * l; }
constexpr int STASH_ROWS = 5 * HROWS + 4 * D + NOUT_MAX;  // 596 (the dZ_4 rows beyond n_out hold zeros)
constexpr int CHUNK = 1 << 19;  // samples per stash chunk: 596 rows x 2^19 x 4 B = 1.24 GB of workspace (288 GB HBM)
constexpr int NPAIRS = 39;                      // 4 layers x 3x3 tiles + last layer 1x3

struct Z33 { float v[33]; };

// timesteps index per-timestep tables: clamped to [0, T-1] wherever a kernel gathers with them (the reference raises IndexError
// outside that range; a kernel cannot, and must not read outside its tables -- a caller's uninitialised buffer included)
__device__ __forceinline__ int64_t clamp_t(int64_t v, int T) { return T <= 0 ? v : (v < 0 ? 0 : (v >= T ? T - 1 : v)); }  // [0..15] tile 0, [16..31] tile 1, [32] = tile 2 reg 0 (feature 64 in the lower half)

// silu and its derivative from the pre-activation
template <int PREC> __device__ __forceinline__ void silu_grad(float z, float* hval, float* dval) {
  float sg;
  if (PREC == SO3X_PREC_F32) sg = sigmoid_f32(z);
  else sg = __builtin_amdgcn_rcpf(1.0f + __expf(-z));
  const float hv = z * sg;
  *hval = hv;
  *dval = sg + hv * (1.0f - sg);  // sigma (1 + z (1 - sigma))
}

// dH = W^T dZ for one layer; dz in the 33-register layout of Z33 (layer-4: only v[0..5] of the lower half)
template <int PREC, int L>
__device__ __forceinline__ void dh_layer(const void* __restrict__ wt, const Z33& dz, f32x16 (&dh)[3], int lane) {
  if constexpr (PREC == SO3X_PREC_F32) {
    const float* w = reinterpret_cast<const float*>(wt);
    constexpr int KS = L < 4 ? 33 : k4<PREC>();
#pragma unroll
    for (int to = 0; to < 3; to++) {
      __builtin_amdgcn_sched_barrier(0);  // bound the LDS-read hoisting to one output tile
      f32x16 a = zero16<PREC>();
#pragma unroll
      for (int ks = 0; ks < KS; ks++) a = mfma_f32(w[(size_t)wt_frag<PREC>(L, to, ks) * 64 + lane], dz.v[ks], a);
      dh[to] = a;
    }
  } else {
    const bf16x8* w = reinterpret_cast<const bf16x8*>(wt);
    constexpr int KS = L < 4 ? 5 : 1;
    bf16x8 b[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ks++) {
#pragma unroll
      for (int j = 0; j < 8; j++) b[ks][j] = (__bf16)(ks == 4 ? (j == 0 ? dz.v[32] : 0.0f) : dz.v[8 * ks + j]);
    }
#pragma unroll
    for (int to = 0; to < 3; to++) {
      __builtin_amdgcn_sched_barrier(0);
      f32x16 a = zero16<PREC>();
#pragma unroll
      for (int ks = 0; ks < KS; ks++) a = mfma_bf16(w[(size_t)wt_frag<PREC>(L, to, ks) * 64 + lane], b[ks], a);
      dh[to] = a;
    }
  }
}

// Stash stores use a wave-uniform 64-bit row base (SGPRs) plus a per-lane 32-bit offset, so the 270 stores of a
// tile cost no VGPR address pairs (per-lane 64-bit addresses made the register allocator spill ~400 VGPRs).
//   off4 = 4 h nc + s  (rows row_of(r, h) = row_of(r, 0) + 4 h),   off1 = h nc + s  (rows base + h)
template <int PREC> struct Stash { using T = float; };      // fp32 mode: exact hand-off
template <> struct Stash<SO3X_PREC_BF16> { using T = __bf16; };  // bf16 mode: half the HBM round trip, bf16 MFMA in K2
template <typename T>
__device__ __forceinline__ void st_row(T* __restrict__ stash, int64_t nc, int row0, unsigned off, float v) {
  // non-temporal: the fp32 / unbounded-t stash is a once-through stream beside the images and tables that K1 re-reads from
  // L2 every round (fp32 training step at 2^19 samples 2.19 -> 2.03 ms)
  __builtin_nontemporal_store((T)v, stash + (int64_t)row0 * nc + off);
}

// stash row of feature-layout register q (Z33 order) for lane half h: rows 0..63 for q < 32; q == 32: feature 64 (h = 0)
__device__ __forceinline__ int z33_row0(int q) { return q < 32 ? 32 * (q >> 4) + row_of(q & 15, 0) : 64; }

// ---------------------------------------------------------------------------------------
// K1.  LDS holds the forward image and the transposed image.  In bf16 both fit (65 + 48 KB)
// and the waves run free; in fp32 (110 + 78 KB > 160 KB) the block swaps the two images
// around a barrier once per tile round (4 tiles per block per round).
// Pre-activations are kept as 33 live registers per layer (tiles 0,1 + feature 64).
// ---------------------------------------------------------------------------------------
template <int PREC> __host__ __device__ constexpr bool swap_images() { return PREC == SO3X_PREC_F32; }
template <int PREC, int VAR> __host__ __device__ constexpr int stage_lds_bytes() {
  return swap_images<PREC>() ? (image_bytes<PREC, VAR>() > wt_bytes<PREC>() ? image_bytes<PREC, VAR>() : wt_bytes<PREC>())
                             : image_bytes<PREC, VAR>() + wt_bytes<PREC>();
}


__device__ __forceinline__ void keep(const f32x16 (&acc)[3], Z33& z) {
#pragma unroll
  for (int r = 0; r < 16; r++) { z.v[r] = acc[0][r]; z.v[16 + r] = acc[1][r]; }
  z.v[32] = acc[2][0];
}

template <int PREC>
__device__ __forceinline__ void activate_z(const Z33& z, Tile<PREC>& out, int h) {
  f32x16 a[3];
#pragma unroll
  for (int r = 0; r < 16; r++) { a[0][r] = z.v[r]; a[1][r] = z.v[16 + r]; a[2][r] = 0.0f; }
  a[2][0] = z.v[32];
  activate<PREC>(a, out, h);
}

template <int PREC, int VAR>
__global__ void __launch_bounds__(256, 1)
k_bwd_stage(const void* __restrict__ gimg, const void* __restrict__ gwt, const float* __restrict__ beff_tab,
            const float* __restrict__ emb_tab, const float* __restrict__ R, const int64_t* __restrict__ t,
            int64_t t_stride, const float* __restrict__ dout, Freqs fr, typename Stash<PREC>::T* __restrict__ stash,
            int64_t nc /*samples in this chunk*/, int nout, int T /*rows of the per-timestep tables; 0 = none*/) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  constexpr bool SWAP = swap_images<PREC>();
  constexpr int FB = frag_bytes<PREC>();
  constexpr int IMG = image_bytes<PREC, VAR>();
  const char* wt_lds = SWAP ? lds : lds + IMG;
  load_image(gimg, lds, IMG);
  if (!SWAP) load_image(gwt, lds + IMG, wt_bytes<PREC>());
  __syncthreads();
  const int lane = threadIdx.x & 63, col = lane & 31, h = lane >> 5;
  const int64_t ntiles = (nc + 31) / 32;
  const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
  const int64_t rounds = (ntiles + nwaves - 1) / nwaves;  // uniform trip count: the swap barriers are block-wide
  for (int64_t rd = 0; rd < rounds; rd++) {
    const int64_t tile = rd * nwaves + wave;
    const bool active = tile < ntiles;
    if (SWAP && rd > 0) {
      __syncthreads();
      load_image(gimg, lds, IMG);
      __syncthreads();
    }
    const int64_t s = tile * 32 + col;
    const bool live = active && s < nc;
    const int64_t sc = live ? s : nc - 1;
    const unsigned off1 = (unsigned)(h * nc + s), off4 = (unsigned)(4 * h * nc + s);
    Z33 z[4];
    float dd[NOUT_MAX] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (active) {
      float x[9];
#pragma unroll
      for (int j = 0; j < 9; j++) x[j] = R[sc * 9 + j];
      const int64_t tt = VAR == GATHER ? clamp_t(t[sc * t_stride], T) : t[sc * t_stride];
#pragma unroll
      for (int k = 0; k < NOUT_MAX; k++) dd[k] = k < nout ? dout[sc * nout + k] : 0.0f;
      // ---- layer-0 input rows of the stash: [0..8] R, [9] one, [10..65] emb (canonical order);
      //      the two lanes of a sample split the rows: lane half h writes rows 2 j + h
      if (live) {
#pragma unroll
        for (int j = 0; j < 5; j++) {
          const float lo = x[2 * j], hi = (2 * j + 1 < 9) ? x[(2 * j + 1) % 9] : 1.0f;
          st_row(stash, nc, H_BASE(0) + 2 * j, off1, h ? hi : lo);
        }
#pragma unroll 1
        for (int e = 0; e < NEMB; e += 2)
          st_row(stash, nc, H_BASE(0) + 10 + e, off1,
                 VAR == GATHER ? emb_tab[(size_t)tt * NEMB + e + h] : emb_value(tt, e + h, fr));
      }
      // ---- forward, keeping the pre-activations
      f32x16 acc[3];
      Tile<PREC> cur;
      if constexpr (VAR == GATHER) layer0_chain<PREC, 0>(lds, beff_tab + (size_t)tt * 96, x, acc, lane);
      else layer0_full<PREC>(lds, x, tt, fr, acc, lane);
      keep(acc, z[0]);
#pragma unroll
      for (int l = 1; l < 4; l++) {
        activate_z<PREC>(z[l - 1], cur, h);
        hidden_layer<PREC, 3>(lds + (size_t)frag_hidden<PREC, VAR>(l) * FB, cur, acc, lane);
        keep(acc, z[l]);
      }
    }
    if (SWAP) {
      __syncthreads();
      load_image(gwt, lds, wt_bytes<PREC>());
      __syncthreads();
    }
    if (active) {
      // ---- backward
      Z33 dz;
      f32x16 dh[3];
#pragma unroll
      for (int q = 0; q < 33; q++) dz.v[q] = 0.0f;
      if (h == 0) {
#pragma unroll
        for (int k = 0; k < NOUT_MAX; k++) {
          dz.v[k] = dd[k];                                          // K slots 0..5 of the lower half (head_of_row)
          if (live) st_row(stash, nc, DZ_BASE(4) + k, off1, dd[k]);  // h == 0 here: off1 == s
        }
      }
      dh_layer<PREC, 4>(wt_lds, dz, dh, lane);
#pragma unroll
      for (int l = 3; l >= 0; l--) {
        // H_{l+1} = silu(Z_l) is the input of layer l+1; dZ_l = dH . silu'(Z_l)
#pragma unroll
        for (int q = 0; q < 33; q++) {
          float a, d;
          silu_grad<PREC>(z[l].v[q], &a, &d);
          const float g = (q < 16 ? dh[0][q] : (q < 32 ? dh[1][q - 16] : dh[2][0])) * d;
          const bool own = q < 32 || h == 0;  // upper half of tile 2 / reg 0 is the constant-one row: no gradient
          dz.v[q] = own ? g : 0.0f;
          if (live) {
            if (q < 32) {
              st_row(stash, nc, H_BASE(l + 1) + z33_row0(q), off4, a);
              st_row(stash, nc, DZ_BASE(l) + z33_row0(q), off4, g);
            } else {  // lower half: feature 64; upper half: the constant-one row 65 (bias carrier), no dZ row
              st_row(stash, nc, H_BASE(l + 1) + 64, off1, h ? 1.0f : a);
              if (h == 0) st_row(stash, nc, DZ_BASE(l) + 64, off1, g);
            }
          }
        }
        if (l == 3) dh_layer<PREC, 3>(wt_lds, dz, dh, lane);
        if (l == 2) dh_layer<PREC, 2>(wt_lds, dz, dh, lane);
        if (l == 1) dh_layer<PREC, 1>(wt_lds, dz, dh, lane);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------
// K2: dW tiles.  Pair p -> (layer, To = out-feature tile of dZ_l, Ti = in-feature tile of H_l)
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ void pair_of(int p, int* l, int* to, int* ti) {
  if (p < 36) { *l = p / 9; *to = (p % 9) / 3; *ti = p % 3; }
  else { *l = 4; *to = 0; *ti = p - 36; }
}

__device__ __forceinline__ void write_slab_impl(float* __restrict__ slabs, const f32x16 (&acc)[5], int wid, int i, int h, int nout);
__device__ __forceinline__ void write_slab(float* __restrict__ slabs, const f32x16 (&acc)[5], int wid, int i, int h, int nout) {
  write_slab_impl(slabs, acc, wid, i, h, nout);
}

constexpr int LROW = 33;  // padded LDS row (floats): fragment reads of stride-LROW rows are conflict-free

// Double-buffered: the next 32-sample slice travels global -> registers while this one's MFMAs run, and lands in the other
// LDS buffer afterwards (one barrier per slice).  Single-buffered, the 76 KB load and the ~10k-cycle MFMA phase of a slice
// ran back to back: 0.89 ms at 2^19 samples, 35 % of the fp32 MFMA rate.
constexpr int DW_PIECES = (STASH_ROWS * 8 + 511) / 512;  // 16-byte pieces of a slice per thread

__global__ void __launch_bounds__(512, 1)
k_bwd_dw(const float* __restrict__ stash, int64_t nc, float* __restrict__ slabs, int n_out) {
  extern __shared__ __attribute__((aligned(16))) float sm[];  // 2 x [STASH_ROWS][LROW]
  constexpr int BUF = STASH_ROWS * LROW;
  const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int i = lane & 31, h = lane >> 5;
  f32x16 acc[5];
#pragma unroll
  for (int k = 0; k < 5; k++) acc[k] = zero16<0>();
  const int64_t nsub = (nc + 31) / 32;
  typedef float sf4 __attribute__((ext_vector_type(4)));
  sf4 pv[DW_PIECES];
  // [593 rows][32 samples]: 8 lanes x 16 B per row, coalesced along the sample axis
  auto fetch = [&](int64_t sub) {
    const int64_t s0 = sub * 32;
#pragma unroll
    for (int q = 0; q < DW_PIECES; q++) {
      const int e = threadIdx.x + 512 * q;
      sf4 v = {0.f, 0.f, 0.f, 0.f};
      if (e < STASH_ROWS * 8) {
        const int row = e >> 3, c4 = (e & 7) * 4;
        const float* src = stash + (int64_t)row * nc + s0 + c4;
        if (s0 + c4 + 3 < nc && ((reinterpret_cast<uintptr_t>(src) & 15) == 0)) {
          v = __builtin_nontemporal_load(reinterpret_cast<const sf4*>(src));
        } else {
          v.x = (s0 + c4 + 0 < nc) ? src[0] : 0.0f; v.y = (s0 + c4 + 1 < nc) ? src[1] : 0.0f;
          v.z = (s0 + c4 + 2 < nc) ? src[2] : 0.0f; v.w = (s0 + c4 + 3 < nc) ? src[3] : 0.0f;
        }
      }
      pv[q] = v;
    }
  };
  auto commit = [&](float* buf) {
#pragma unroll
    for (int q = 0; q < DW_PIECES; q++) {
      const int e = threadIdx.x + 512 * q;
      if (e < STASH_ROWS * 8) {
        float* d = buf + (e >> 3) * LROW + (e & 7) * 4;
        d[0] = pv[q].x; d[1] = pv[q].y; d[2] = pv[q].z; d[3] = pv[q].w;
      }
    }
  };
  int64_t sub = blockIdx.x;
  if (sub < nsub) { fetch(sub); commit(sm); }
  __syncthreads();
  int cur = 0;
  for (; sub < nsub; sub += gridDim.x) {
    const int64_t nxt = sub + gridDim.x;
    if (nxt < nsub) fetch(nxt);
    const float* buf = sm + cur * BUF;
    // (tile-outer order on purpose: advancing the wave's five tiles together -- five independent accumulator chains --
    //  measured 7 % slower: ten LDS reads per step leave no room to run ahead)
#pragma unroll
    for (int k = 0; k < 5; k++) {
      const int p = wid + 8 * k;
      if (p < NPAIRS) {
        int l, to, ti;
        pair_of(p, &l, &to, &ti);
        const int nout = (l < 4 ? D : NOUT_MAX) - 32 * to;   // valid dZ rows in this tile
        const int nin = HROWS - 32 * ti;              // valid H rows in this tile
        const bool va = i < nout, vb = i < nin;
        const float* pa = buf + (DZ_BASE(l) + 32 * to + (va ? i : 0)) * LROW + h;
        const float* pb = buf + (H_BASE(l) + 32 * ti + (vb ? i : 0)) * LROW + h;
        f32x16 a = acc[k];
#pragma unroll
        for (int m = 0; m < 16; m++) {
          const float fa = va ? pa[2 * m] : 0.0f;
          const float fb = vb ? pb[2 * m] : 0.0f;
          a = mfma_f32(fa, fb, a);
        }
        acc[k] = a;
      }
    }
    if (nxt < nsub) commit(sm + (cur ^ 1) * BUF);
    __syncthreads();  // the other buffer is complete; everyone is done reading this one
    cur ^= 1;
  }
  write_slab(slabs, acc, wid, i, h, n_out);
}

// this block's partial dparams slab (every entry is owned by exactly one lane of one wave)
__device__ __forceinline__ void write_slab_impl(float* __restrict__ slabs, const f32x16 (&acc)[5], int wid, int i, int h, int nout) {
  float* slab = slabs + (size_t)blockIdx.x * NPARAMS_MAX;
#pragma unroll
  for (int k = 0; k < 5; k++) {
    const int p = wid + 8 * k;
    if (p < NPAIRS) {
      int l, to, ti;
      pair_of(p, &l, &to, &ti);
      const int in_ext = 32 * ti + i;  // D-layout: lane column = H row (in), register rows = dZ row (out)
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int out = 32 * to + row_of(r, h);
        if (out >= (l < 4 ? D : nout) || in_ext >= HROWS) continue;
        int col;  // column of the layer's weight, or -2 = bias
        if (l == 0) col = in_ext < 9 ? in_ext : (in_ext == 9 ? -2 : in_ext - 1);  // [R(9), one, emb(56)] -> cols 0..8, bias, 9..64
        else col = in_ext < D ? in_ext : -2;
        const int base = l * LAYER_STRIDE;
        const int idx = col >= 0 ? base + out * D + col : base + (l < 4 ? D : nout) * D + out;
        slab[idx] = acc[k][r];
      }
    }
  }
}

// bf16 stash variant: rows of 32 bf16 (64 B) staged at an 80-B pitch (16-B aligned, ds_read_b128 conflict-free),
// two v_mfma_f32_32x32x16_bf16 k-steps per 32-sample slice (k = sample index), fp32 accumulation.
constexpr int LROW16 = 40;  // bf16 elements per padded LDS row

__global__ void __launch_bounds__(512, 1)
k_bwd_dw_bf16(const __bf16* __restrict__ stash, int64_t nc, float* __restrict__ slabs, int n_out) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  __bf16* sb = reinterpret_cast<__bf16*>(sm);  // [STASH_ROWS][LROW16]
  const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int i = lane & 31, h = lane >> 5;
  f32x16 acc[5];
#pragma unroll
  for (int k = 0; k < 5; k++) acc[k] = zero16<0>();
  bf16x8 zero8;
#pragma unroll
  for (int j = 0; j < 8; j++) zero8[j] = (__bf16)0.0f;
  const int64_t nsub = (nc + 31) / 32;
  for (int64_t sub = blockIdx.x; sub < nsub; sub += gridDim.x) {
    const int64_t s0 = sub * 32;
    __syncthreads();
    for (int e = threadIdx.x; e < STASH_ROWS * 4; e += blockDim.x) {  // 4 lanes x 16 B per row
      const int row = e >> 2, c8 = (e & 3) * 8;
      const __bf16* src = stash + (int64_t)row * nc + s0 + c8;
      bf16x8 v;
      if (s0 + c8 + 7 < nc && ((reinterpret_cast<uintptr_t>(src) & 15) == 0)) {
        v = *reinterpret_cast<const bf16x8*>(src);
      } else {
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = (s0 + c8 + u < nc) ? src[u] : (__bf16)0.0f;
      }
      *reinterpret_cast<bf16x8*>(sb + row * LROW16 + c8) = v;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 5; k++) {
      const int p = wid + 8 * k;
      if (p < NPAIRS) {
        int l, to, ti;
        pair_of(p, &l, &to, &ti);
        const int nout = (l < 4 ? D : NOUT_MAX) - 32 * to;
        const int nin = HROWS - 32 * ti;
        const bool va = i < nout, vb = i < nin;
        const __bf16* pa = sb + (DZ_BASE(l) + 32 * to + (va ? i : 0)) * LROW16 + 8 * h;
        const __bf16* pb = sb + (H_BASE(l) + 32 * ti + (vb ? i : 0)) * LROW16 + 8 * h;
        f32x16 a = acc[k];
#pragma unroll
        for (int m = 0; m < 2; m++) {  // k-step m: samples 16 m + 8 h + j
          const bf16x8 fa = va ? *reinterpret_cast<const bf16x8*>(pa + 16 * m) : zero8;
          const bf16x8 fb = vb ? *reinterpret_cast<const bf16x8*>(pb + 16 * m) : zero8;
          a = mfma_bf16(fa, fb, a);
        }
        acc[k] = a;
      }
    }
  }
  write_slab(slabs, acc, wid, i, h, n_out);
}

// Slab reduction.  A block = RED_COLS float4 columns (four consecutive parameters each: one 16-byte load per slab) x RED_GROUPS slab
// groups; every thread's loads are independent and in flight together -- the kernel is a latency chain, not a bandwidth problem
// (18 MB) -- and the summation order is FIXED: per element four running sums over the slabs of its group (b = g, g + 16, ...),
// combined as (s0 + s1) + (s2 + s3), then the 16 groups in order.  (Round 4: 16-byte columns and 256-thread blocks instead of
// 4-byte ones and 1,024 -- a quarter of the load instructions; the order per element, hence every bit, is round 2's.)
constexpr int RED_COLS = 16, RED_GROUPS = 16, RED_THREADS = RED_COLS * RED_GROUPS;
__host__ __device__ constexpr int red_blocks(int np) { return ((np + 3) / 4 + RED_COLS - 1) / RED_COLS; }
static_assert(NPARAMS_MAX % 4 == 0, "slabs are read in 16-byte columns");
__device__ __forceinline__ float4 slab_sum4(const float* __restrict__ slabs, int nslabs, int np, float4 (*part)[RED_COLS + 1], int* col_out) {
  const int c = threadIdx.x % RED_COLS, g = threadIdx.x / RED_COLS;
  const int col = blockIdx.x * RED_COLS + c;
  *col_out = col;
  float4 s = {0.f, 0.f, 0.f, 0.f};
  if (4 * col < np) {
    const float4* base = reinterpret_cast<const float4*>(slabs) + col;
    constexpr size_t STRIDE4 = NPARAMS_MAX / 4;
    float4 s0 = s, s1 = s, s2 = s, s3 = s;
    auto add = [](float4& a, const float4 v) { a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w; };
    int b = g;
    for (; b + 3 * RED_GROUPS < nslabs; b += 4 * RED_GROUPS) {
      add(s0, base[(size_t)b * STRIDE4]);
      add(s1, base[(size_t)(b + RED_GROUPS) * STRIDE4]);
      add(s2, base[(size_t)(b + 2 * RED_GROUPS) * STRIDE4]);
      add(s3, base[(size_t)(b + 3 * RED_GROUPS) * STRIDE4]);
    }
    for (; b < nslabs; b += RED_GROUPS) add(s0, base[(size_t)b * STRIDE4]);
    s = float4{(s0.x + s1.x) + (s2.x + s3.x), (s0.y + s1.y) + (s2.y + s3.y), (s0.z + s1.z) + (s2.z + s3.z), (s0.w + s1.w) + (s2.w + s3.w)};
  }
  part[g][c] = s;
  __syncthreads();
  float4 t = {0.f, 0.f, 0.f, 0.f};
  if (g == 0 && 4 * col < np) {
#pragma unroll
    for (int k = 0; k < RED_GROUPS; k++) { const float4 v = part[k][c]; t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w; }
  }
  return t;   // valid in the threads of group 0
}

__global__ void __launch_bounds__(RED_THREADS)
k_bwd_reduce(const float* __restrict__ slabs, int nslabs, float* __restrict__ dparams, int accumulate, int np,
             const float* __restrict__ gscale = nullptr, const float* __restrict__ status = nullptr) {
  __shared__ float4 part[RED_GROUPS][RED_COLS + 1];
  int col;
  const float4 t4 = slab_sum4(slabs, nslabs, np, part, &col);
  // status (the training step's loss): not finite = a hand-shake of the step's kernel gave up and its slabs are partial.  The
  // gradient then is all NaN -- loud, and so3x_adam_step skips a non-finite gradient.
  const bool bad = status != nullptr && !(fabsf(status[0]) <= 3.0e38f);
  if (threadIdx.x / RED_COLS == 0) {
    const float tv[4] = {t4.x, t4.y, t4.z, t4.w};
    const float gs = gscale ? gscale[0] : 1.0f;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int idx = 4 * col + k;
      if (idx < np) {
        const float t = bad ? __builtin_nanf("") : (gscale ? tv[k] * gs : tv[k]);
        dparams[idx] = accumulate ? dparams[idx] + t : t;
      }
    }
  }
}

// The same reduction with torch.optim.Adam's update of the reduced parameters in its epilogue (so3x_optim.hip's k_adam, term for
// term): in a single process nothing sits between the slab reduction and the optimizer, and a 17,358-element update is all launch
// latency (5 us of a 0.235 ms step).  Every block reads the step count before any block advances it (the last one to arrive does).
struct AdamArgs { float* p; float* m; float* v; float* step; unsigned* ticket; float lr, beta1, beta2, eps, weight_decay, grad_scale; };
__global__ void __launch_bounds__(RED_THREADS)
k_bwd_reduce_adam(const float* __restrict__ slabs, int nslabs, float* __restrict__ dparams, int np, const float* __restrict__ gscale,
                  AdamArgs ad, const float* __restrict__ status) {
  __shared__ float4 part[RED_GROUPS][RED_COLS + 1];
  __shared__ float sc[2];
  if (threadIdx.x == RED_THREADS - 1) {  // (a thread of the last group: the scalars are ready when the partial sums are)
    const double k = (double)ad.step[0] + 1.0;
    const double bc1 = 1.0 - pow((double)ad.beta1, k), bc2 = 1.0 - pow((double)ad.beta2, k);
    sc[0] = (float)(-(double)ad.lr / bc1);
    sc[1] = (float)sqrt(bc2);
  }
  int col;
  const float4 t4 = slab_sum4(slabs, nslabs, np, part, &col);
  // a step whose kernel gave up on a hand-shake (loss not finite): gradient NaN, parameters / moments / step count untouched
  const bool bad = status != nullptr && !(fabsf(status[0]) <= 3.0e38f);
  if (threadIdx.x / RED_COLS == 0) {
    const float tv[4] = {t4.x, t4.y, t4.z, t4.w};
    const float gs = gscale ? gscale[0] : 1.0f;
    const float neg_step_size = sc[0], bc2_sqrt = sc[1];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int idx = 4 * col + k;
      if (idx >= np) continue;
      const float t = gscale ? tv[k] * gs : tv[k];
      dparams[idx] = bad ? __builtin_nanf("") : t;
      if (bad) continue;
      float gi = t * ad.grad_scale;
      const float pi = ad.p[idx];
      if (ad.weight_decay != 0.0f) gi = fmaf(ad.weight_decay, pi, gi);
      float mi = ad.m[idx], vi = ad.v[idx];
      mi = mi + (1.0f - ad.beta1) * (gi - mi);
      vi = vi * ad.beta2 + (1.0f - ad.beta2) * gi * gi;
      const float denom = sqrtf(vi) / bc2_sqrt + ad.eps;
      ad.p[idx] = pi + neg_step_size * (mi / denom);
      ad.m[idx] = mi;
      ad.v[idx] = vi;
    }
  }
  if (threadIdx.x == RED_THREADS - 1) {
    const unsigned mine = __hip_atomic_fetch_add(ad.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (mine == gridDim.x - 1) {
      __hip_atomic_store(ad.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (!bad) ad.step[0] = ad.step[0] + 1.0f;
    }
  }
}

// ---------------------------------------------------------------------------------------
// Fused backward (bf16, bounded timesteps): K1 and K2 in ONE kernel, no HBM stash.
//   * a block = 4 waves = 128 samples per round; every wave recomputes the forward of its 32-sample tile
//     (pre-activations in registers) and walks the layers backwards;
//   * per layer, each wave drops its dZ_l and H_l tiles into its own 12 KB LDS image
//     [32 samples][192 features] with 8-byte stores straight from the accumulator layout (4 consecutive
//     feature rows per store), the block synchronises, and the 39 dW tiles -- owned 10 / 10 / 10 / 9 by
//     the four waves, accumulators persistent in registers for the whole launch -- are updated with
//     v_mfma_f32_32x32x16_bf16 whose operands come from ds_read_b64_tr_b16 (hardware transposed read:
//     the contraction index of dW = dZ H^T is the SAMPLE, which is the lane index of the producers);
//   * the image is XOR-swizzled so that the 8-byte stores (16-lane groups, 32-dword banking) and the
//     transposed reads (32-lane halves, 64-dword banking) are both bank-conflict free:
//         chunk' = chunk ^ ((row & 7) | ((((row >> 1) ^ (row >> 3)) & 1) << 3)),  8-byte chunks, 384-byte rows.
// LDS: forward image 53 KB + transposed image 48 KB + 4 x 12 KB = 149 KB (one block per CU).
// ---------------------------------------------------------------------------------------
// pre-activations of the fused kernel are parked as packed f16 (66 -> 17 VGPRs per layer would be 33 fp32):
// 11 significant bits on O(1..10) values, well inside the bf16 path's tolerance, and it is what lets the
// 160 persistent dW accumulator registers + 4 layers of Z fit the 512-register budget without scratch.
struct Z33h {
  typedef _Float16 h2 __attribute__((ext_vector_type(2)));
  h2 p[17];
  __device__ __forceinline__ float get(int q) const { return (float)p[q >> 1][q & 1]; }
};
__device__ __forceinline__ void keep_h(const f32x16 (&acc)[3], Z33h& z) {
#pragma unroll
  for (int r = 0; r < 16; r += 2) {
    z.p[r >> 1] = Z33h::h2{(_Float16)acc[0][r], (_Float16)acc[0][r + 1]};
    z.p[8 + (r >> 1)] = Z33h::h2{(_Float16)acc[1][r], (_Float16)acc[1][r + 1]};
  }
  z.p[16] = Z33h::h2{(_Float16)acc[2][0], (_Float16)0.0f};
}
template <int PREC>
__device__ __forceinline__ void activate_zh(const Z33h& z, Tile<PREC>& out, int h) {
  f32x16 a[3];
#pragma unroll
  for (int r = 0; r < 16; r++) { a[0][r] = z.get(r); a[1][r] = z.get(16 + r); a[2][r] = 0.0f; }
  a[2][0] = z.get(32);
  activate<PREC>(a, out, h);
}

// Pre-activation stash between the training forward (k_mlp_fwd_stash) and k_bwd_fused: the four Z33h of a wave's
// 32-sample tile as they sit in registers, 17 dwords per lane and layer = four fully coalesced 1-KiB chunks and one
// 256-B row per layer, layer after layer, so that the backward can fetch them one layer ahead of their use.
// 17 KiB per tile = 544 B per sample.
// (Non-temporal hints on this stash were measured and left out: NT stores make the training forward 16 % faster but the
//  backward, which then finds nothing in the memory-side cache, 15 % slower; NT loads cost the backward 12 %.)
constexpr size_t ZSTASH_TILE = 17 * 1024;
constexpr size_t ZSTASH_LAYER = 17 * 256;  // one layer of a tile: 4 chunks of [lane][16 B], then [lane][4 B]
// The stash is written once and read once, 285 MB per 2^19 samples -- more than the Infinity Cache: non-temporal stores took
// the training forward from 87 to 70 us (default-policy lines were being written back behind the kernel's own compute).

__device__ __forceinline__ void zstash_store_layer(char* tile_base, int lane, int l, const Z33h& z) {
  char* lb = tile_base + l * ZSTASH_LAYER;
  uint4* o = reinterpret_cast<uint4*>(lb) + lane;
  typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
#pragma unroll
  for (int c = 0; c < 4; c++) {
    const u32x4_t v = {__builtin_bit_cast(uint32_t, z.p[4 * c]), __builtin_bit_cast(uint32_t, z.p[4 * c + 1]),
                       __builtin_bit_cast(uint32_t, z.p[4 * c + 2]), __builtin_bit_cast(uint32_t, z.p[4 * c + 3])};
    __builtin_nontemporal_store(v, reinterpret_cast<u32x4_t*>(o + c * 64));
  }
  __builtin_nontemporal_store(__builtin_bit_cast(uint32_t, z.p[16]), reinterpret_cast<uint32_t*>(lb + 4096) + lane);
}
// (uint4, a struct of four words, on purpose: with an ext-vector load and `__builtin_bit_cast(h2, v[i])` of its ELEMENTS this
//  hipcc builds a backward whose gradients are 30 % off -- seen twice, tools/ab/cmp_train_grad.py is the check)
__device__ __forceinline__ void zstash_load_layer(const char* tile_base, int lane, int l, Z33h& z) {
  const char* lb = tile_base + l * ZSTASH_LAYER;
  const uint4* in = reinterpret_cast<const uint4*>(lb) + lane;
#pragma unroll
  for (int c = 0; c < 4; c++) {
    const uint4 v = in[c * 64];
    z.p[4 * c] = __builtin_bit_cast(Z33h::h2, v.x); z.p[4 * c + 1] = __builtin_bit_cast(Z33h::h2, v.y);
    z.p[4 * c + 2] = __builtin_bit_cast(Z33h::h2, v.z); z.p[4 * c + 3] = __builtin_bit_cast(Z33h::h2, v.w);
  }
  z.p[16] = __builtin_bit_cast(Z33h::h2, reinterpret_cast<const uint32_t*>(lb + 4096)[lane]);
}

// Training forward (bf16 operands, per-timestep tables): the network output AND the stash above, so that the backward
// does not run the forward again (that recompute was half of k_bwd_fused's time).  Same arithmetic as the recompute it
// replaces: GATHER_T image (table SiLU; stash_forward_tile above), pre-activations parked as f16.
// LOSS: the MSE of p_losses rides in the epilogue -- the lane that holds a sample's three outputs reads its target, writes
// d loss / d out and adds to a running sum of squares; per-block sums are combined by the last block to arrive, in block
// order (deterministic) -- so the loss, its gradient and the step's counter increment cost no launch of their own.
// The training forward on one 32-sample tile (bf16, GATHER_T image: table SiLU, so3x_mlp.hpp).  The hidden layers' MFMAs emit
// the table coordinate u = 16 z + 127.5; the activation is the chain kernel's table lookup (3 vector instructions instead of
// ~9 with two transcendentals -- the forward was vector-issue-bound at 57 % busy, its stash stores 28 of its 88 us), and the
// pre-activation the backward needs is recovered as z = u / 16 - 127.5 / 16 (exact to 1e-6, far inside the stash's f16).
__device__ __forceinline__ void keep_h_folded(const f32x16 (&u)[3], Z33h& z) {
  constexpr float ks = 1.0f / kTabC, ko = -kTabD / kTabC;
#pragma unroll
  for (int r = 0; r < 16; r += 2) {
    z.p[r >> 1] = Z33h::h2{(_Float16)fmaf(u[0][r], ks, ko), (_Float16)fmaf(u[0][r + 1], ks, ko)};
    z.p[8 + (r >> 1)] = Z33h::h2{(_Float16)fmaf(u[1][r], ks, ko), (_Float16)fmaf(u[1][r + 1], ks, ko)};
  }
  z.p[16] = Z33h::h2{(_Float16)fmaf(u[2][0], ks, ko), (_Float16)0.0f};
}
// Every layer's pre-activations leave for the stash as soon as they exist (ztile: the tile's 17 KB, nullptr = a padding
// tile): four 4-KB bursts spread over the tile's compute instead of one 17-KB burst behind it -- the stash is 285 MB per 2^19
// samples, written at the HBM's store rate, and waves that all store at once serialise with their compute.
template <int XSRC, bool WIDE = false>
__device__ __forceinline__ void stash_forward_tile(const char* __restrict__ lds, const float* __restrict__ beff_row, const float* x,
                                                   char* __restrict__ ztile, f32x16 (&last)[1], int lane, uint32_t lt = 0) {
  constexpr int PREC = SO3X_PREC_BF16, VAR = GATHER_T, FB = frag_bytes<PREC>();
  const int h = lane >> 5;
  const char* tab = lds + (size_t)n_frags<PREC, VAR>() * FB;
  f32x16 a3[3];
  Tile<PREC> cur;
  Z33h z;
  layer0_chain<PREC, XSRC>(lds, beff_row, x, a3, lane);
  keep_h_folded(a3, z);
  if (ztile) zstash_store_layer(ztile, lane, 0, z);
#pragma unroll
  for (int l = 1; l < 4; l++) {
    activate_bf16<true, WIDE>(a3, cur, h, tab, lt);
    hidden_layer<PREC, 3>(lds + (size_t)frag_hidden<PREC, VAR>(l) * FB, cur, a3, lane);
    keep_h_folded(a3, z);
    if (ztile) zstash_store_layer(ztile, lane, l, z);
  }
  activate_bf16<true, WIDE>(a3, cur, h, tab, lt);
  hidden_layer<PREC, 1>(lds + (size_t)frag_last<PREC, VAR>() * FB, cur, last, lane);
}

// 8 waves per workgroup, two workgroups per CU (the 57 KB image twice): with each layer's pre-activations stored as soon as
// they exist the kernel needs 124 registers -- four waves per SIMD.
constexpr int kFwdStashThreads = 512;
static_assert(kFwdStashThreads == 512, "the loss epilogue sums eight wave partials");
template <int PREC, bool LOSS>
__global__ void __launch_bounds__(kFwdStashThreads, 2)
k_mlp_fwd_stash(const void* __restrict__ gimg, const float* __restrict__ beff_tab, const float* __restrict__ R,
                const int64_t* __restrict__ t, int64_t t_stride, float* __restrict__ out, char* __restrict__ zstash, int64_t n, int nout,
                LossArgs la, int T) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  static_assert(PREC == SO3X_PREC_BF16, "the training forward is the bf16 path");
  constexpr int VAR = GATHER_T;
  load_image(gimg, lds, image_bytes<PREC, VAR>());
  __syncthreads();
  const int lane = threadIdx.x & 63, col = lane & 31, h = lane >> 5;
  const int64_t ntiles = (n + 31) / 32;
  const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
  float sq = 0.0f;
  for (int64_t tile = wave; tile < ntiles; tile += nwaves) {
    int64_t idx = tile * 32 + col;
    const bool live = idx < n;
    if (!live) idx = n - 1;
    float x[9];
    load_rot9(R, idx, x);
    const int64_t tt = clamp_t(t[idx * t_stride], T);
    f32x16 last[1];
    stash_forward_tile<0>(lds, beff_tab + (size_t)tt * 96, x, zstash + (size_t)tile * ZSTASH_TILE, last, lane);
    if (live && h == 0) {  // head outputs 0..5 = regs 0..5 of the lower half (head_of_row)
      if (out) {
        out[idx * nout] = last[0][0]; out[idx * nout + 1] = last[0][1]; out[idx * nout + 2] = last[0][2];
        if (nout == 6) { out[idx * 6 + 3] = last[0][3]; out[idx * 6 + 4] = last[0][4]; out[idx * 6 + 5] = last[0][5]; }
      }
      if constexpr (LOSS) {
        const float d0 = last[0][0] - la.target[idx * 3], d1 = last[0][1] - la.target[idx * 3 + 1], d2 = last[0][2] - la.target[idx * 3 + 2];
        la.dout[idx * 3] = d0 * la.dscale; la.dout[idx * 3 + 1] = d1 * la.dscale; la.dout[idx * 3 + 2] = d2 * la.dscale;
        sq += d0 * d0 + d1 * d1 + d2 * d2;
      }
    }
  }
  if constexpr (LOSS) {
    __shared__ double wsum[kFwdStashThreads / 64];
    __shared__ int is_last;
    double v = (double)sq;
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    if (lane == 0) wsum[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
      const double bs = ((wsum[0] + wsum[1]) + (wsum[2] + wsum[3])) + ((wsum[4] + wsum[5]) + (wsum[6] + wsum[7]));
      __hip_atomic_store(la.partial + blockIdx.x, bs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      is_last = last_block_arrives(la.ticket) ? 1 : 0;
    }
    __syncthreads();  // the other waves of the last block read the partials only behind this barrier
    if (is_last) {
      // every block's partial, one per thread (the grid has at most 512 blocks), summed in a fixed tree: deterministic
      // whichever block arrives last, and 2 loads deep instead of a 512-long chain of L2 round trips
      double a = 0.0;
      for (unsigned b = threadIdx.x; b < gridDim.x; b += kFwdStashThreads) a += __hip_atomic_load(la.partial + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
      for (int m = 32; m >= 1; m >>= 1) a += __shfl_xor(a, m);
      if (lane == 0) wsum[threadIdx.x >> 6] = a;
      __syncthreads();
      if (threadIdx.x == 0) {
        la.loss[0] = (float)((((wsum[0] + wsum[1]) + (wsum[2] + wsum[3])) + ((wsum[4] + wsum[5]) + (wsum[6] + wsum[7]))) * la.inv_count);
      if (la.status) la.status[0] = la.loss[0];
        if (la.rng_counter) la.rng_counter[0] += 1;  // every reader of this step's offset ran in an earlier launch
      }
    }
  }
}


// one pass over a layer's pre-activations: packed H = silu(Z) (with the constant-one row in the upper half of tile 2) and
// the derivative silu'(Z) in fp32.  H of a dead column (a sample index past n) is NOT zeroed: its dZ is an exact zero in
// every layer (dZ_4 = dout * 0, and the chain maps a zero column to a zero column), so whatever finite H it holds -- the
// forward parks finite pre-activations for every column of every tile -- adds exactly nothing to dW = dZ H^T.  That is
// 33 multiplies per pass off the chain wave (9 % of its vector instructions).
// (MASK: the recomputing variant keeps the multiply by the live flag -- without it the register allocator of THAT kernel,
//  which spills already, does worse.)
template <int PREC, bool MASK = false>
__device__ __forceinline__ void silu_pass(const Z33h& z, int h, uint32_t (&ph)[17], float (&dv)[33], float lv = 1.0f) {
#pragma unroll
  for (int r = 0; r < 16; r++) {
    float a0, a1;
    silu_grad<PREC>(z.get(2 * r), &a0, &dv[2 * r]);
    silu_grad<PREC>(z.get(2 * r + 1), &a1, &dv[2 * r + 1]);
    ph[r] = MASK ? pack_bf16x2(a0 * lv, a1 * lv) : pack_bf16x2(a0, a1);
  }
  float a;
  silu_grad<PREC>(z.get(32), &a, &dv[32]);
  ph[16] = MASK ? pack_bf16x2(h ? lv : a * lv, 0.0f) : pack_bf16x2(h ? 1.0f : a, 0.0f);
}
// The dW role of k_bwd_fused for dW wave DWI (0..3): owns pairs p = DWI + 4k of the 39 (dZ tile, H tile) pairs.
// Ownership of the 39 dW tiles by the four dW waves.  In the hidden layers (3 x 3 tiles each) a wave owns a whole ROW
// of tiles -- the three pairs that share the dZ tile `to` -- so the A fragments are read once per row instead of once per
// pair (64 transposed reads per layer instead of 96); the wave that gets no row in layer l is (l + 3) & 3, so every
// wave works in three of the four layers, and the three tiles of the output layer go one each to waves 0..2: 10/10/10/9
// persistent accumulators.  Everything here is compile-time per DWI.
// The dW role of k_bwd_fused for dW wave DWI (0..3).
template <int PREC, int DWI>
__device__ __forceinline__ void dw_role(const char* fimg_all, int64_t rounds, float* __restrict__ slabs, int lane, int nout) {
  const int col = lane & 31, h = lane >> 5;
  f32x16 acc[10];  // [3 slot + ti] for the hidden layers, [9] = the wave's tile of the output layer
#pragma unroll
  for (int k = 0; k < 10; k++) acc[k] = zero16<PREC>();
  FimgReadLane RL = fimg_read_lane(lane);
  for (int64_t rd = 0; rd < rounds; rd++) {
    asm volatile("" : "+v"(RL.off[0][0]), "+v"(RL.off[0][1]), "+v"(RL.off[1][0]), "+v"(RL.off[1][1]));
#pragma unroll
    for (int l = 4; l >= 0; l--) {
      __syncthreads();
      if (l == 4) {
        if (DWI < 3) {
#pragma unroll
          for (int w = 0; w < 4; w++) {
            const char* im = fimg_all + w * FIMG_BYTES;
#pragma unroll
            for (int ks = 0; ks < 2; ks++) acc[9] = mfma_bf16(fimg_frag(im, RL, 0, ks), fimg_frag(im, RL, 96 + 32 * DWI, ks), acc[9]);
          }
        }
      } else if (dw_row(DWI, l) != 3) {
        constexpr int dummy = 0;
        (void)dummy;
        const int to = dw_row(DWI, l), sl = dw_slot(DWI, l);
#pragma unroll
        for (int w = 0; w < 4; w++) {
          const char* im = fimg_all + w * FIMG_BYTES;
#pragma unroll
          for (int ks = 0; ks < 2; ks++) {
            const bf16x8 a = fimg_frag(im, RL, 32 * to, ks);
#pragma unroll
            for (int ti = 0; ti < 3; ti++) acc[3 * sl + ti] = mfma_bf16(a, fimg_frag(im, RL, 96 + 32 * ti, ks), acc[3 * sl + ti]);
          }
        }
      }
      __syncthreads();
    }
  }
  // ---- slab: D-layout lane column = H feature (in), register rows = dZ feature (out)
  float* slab = slabs + (size_t)blockIdx.x * NPARAMS_MAX;
  auto write_tile = [&](const f32x16& a, int l, int to, int ti) {
    const int in_f = 32 * ti + col;
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const int out = l < 4 ? 32 * to + row_of(r, h) : head_of_row(row_of(r, h));
      if (out < 0 || out >= (l < 4 ? D : nout)) continue;
      int pc;  // weight column, -2 = bias, -1 = padding
      if (l == 0) pc = in_f < 9 ? in_f : (in_f == 9 ? -2 : (in_f < 66 ? in_f - 1 : -1));
      else pc = in_f < D ? in_f : (in_f == ONE_ROW ? -2 : -1);
      if (pc == -1) continue;
      const int base = l * LAYER_STRIDE;
      slab[pc >= 0 ? base + out * D + pc : base + (l < 4 ? D : nout) * D + out] = a[r];
    }
  };
#pragma unroll
  for (int l = 0; l < 4; l++) {
    if (dw_row(DWI, l) == 3) continue;
#pragma unroll
    for (int ti = 0; ti < 3; ti++) write_tile(acc[3 * dw_slot(DWI, l) + ti], l, dw_row(DWI, l), ti);
  }
  if (DWI < 3) write_tile(acc[9], 4, 0, DWI);
}

template <int PREC, bool STASHED>
__global__ void __launch_bounds__(512, 2)
k_bwd_fused(const void* __restrict__ gimg, const void* __restrict__ gwt, const float* __restrict__ beff_tab,
            const float* __restrict__ emb_tab, const float* __restrict__ R, const int64_t* __restrict__ t,
            int64_t t_stride, const float* __restrict__ dout, float* __restrict__ slabs, int64_t n,
            const char* __restrict__ zstash, const uint4* __restrict__ h0_tab, int nout, int T) {
  // Wave specialisation: waves 0-3 ("chain" waves) load (STASHED) or recompute the forward's pre-activations and run the dZ chain for one 32-sample
  // tile each; waves 4-7 ("dW" waves) own the 39 dW tiles (10/10/10/9, persistent accumulators; dw_row) and only consume
  // the LDS images.  One chain wave and one dW wave share a SIMD, so the dW MFMAs run under the chain waves'
  // SiLU-derivative VALU work, and neither role needs more than 256 registers.
  static_assert(PREC == SO3X_PREC_BF16, "fused backward is the bf16 path");
  extern __shared__ __attribute__((aligned(16))) char lds[];
  constexpr int VAR = GATHER;
  constexpr int FB = frag_bytes<PREC>();
  constexpr int IMG = image_bytes<PREC, VAR>();
  constexpr int WTB = wt_bytes<PREC>();
  char* wt_lds = lds + IMG;
  char* fimg_all = lds + IMG + WTB;
  if constexpr (!STASHED) load_image(gimg, lds, IMG);  // the forward image is only for the recompute
  load_image(gwt, wt_lds, WTB);
  for (int i = threadIdx.x; i < 4 * FIMG_BYTES / 16; i += blockDim.x) reinterpret_cast<float4*>(fimg_all)[i] = float4{0.f, 0.f, 0.f, 0.f};
  __syncthreads();
  const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), col = lane & 31, h = lane >> 5;
  const int64_t ntiles = (n + 31) / 32;
  const int64_t nchain = (int64_t)gridDim.x * 4;
  const int64_t rounds = (ntiles + nchain - 1) / nchain;  // uniform trip count: the barriers are block-wide
  if (wid < 4) {
    // =============================== chain waves ===============================
    char* my_img = fimg_all + wid * FIMG_BYTES;
    FimgStoreLane SL = fimg_store_lane(col);
    Z33h zb[2];  // STASHED: the two rolling pre-activation buffers
    // STASHED: the round's timestep and its three dout values are fetched a round ahead (tt_pf, dp_pf): phase stamps showed a
    // quarter of a round going by before its first image store -- an HBM round trip for t and dout on the critical path
    // (then the table row addressed by t), with the rotation held in nine registers for five layers on top
    auto sample_of = [&](int64_t rd_, bool* live_) {
      const int64_t tile_ = rd_ * nchain + (int64_t)blockIdx.x * 4 + wid, s_ = tile_ * 32 + col;
      *live_ = tile_ < ntiles && s_ < n;
      return *live_ ? s_ : n - 1;
    };
    int64_t tt_pf = 0;
    float dp_pf[3] = {0.f, 0.f, 0.f};  // (a 6-wide head fetches its other three on the spot)
    if constexpr (STASHED) {
      bool lv0;
      const int64_t s0 = sample_of(0, &lv0);
      tt_pf = clamp_t(t[s0 * t_stride], T);
      if (h == 0) { dp_pf[0] = dout[s0 * nout]; dp_pf[1] = dout[s0 * nout + 1]; dp_pf[2] = dout[s0 * nout + 2]; }
    }
    // Everything that goes into an image is kept as packed bf16 pairs (the exact MFMA operand bits).
    uint32_t pdz[17], ph[17];
    float dnext[33];  // silu'(Z_{l-1}) of the layer about to be differentiated
    // STASHED: the head of a round -- dZ_4 from the prefetched dout, H_4 = silu(Z_3) and silu'(Z_3) from buffer 0, the fetch
    // of z1 -- runs behind the previous round's LAST first barrier, while the dW waves take the layer-0 products (the chain
    // wave had nothing to do there: 2 k cycles per round at that barrier, and as many for this head at the top of the next)
    auto head_of_round = [&](int64_t rd_) {
      bool live_;
      const int64_t sc_ = sample_of(rd_ < rounds ? rd_ : rounds - 1, &live_);
      const float lv_ = live_ ? 1.0f : 0.0f;
      const int64_t tile_ = rd_ * nchain + (int64_t)blockIdx.x * 4 + wid;
      const char* ztile_ = zstash + (size_t)(tile_ < ntiles ? tile_ : ntiles - 1) * ZSTASH_TILE;
#pragma unroll
      for (int r = 0; r < 17; r++) pdz[r] = 0u;
      if (h == 0) {  // dZ_4 = dout in K slots / image columns 0..5 of the lower half (head_of_row)
        pdz[0] = pack_bf16x2(dp_pf[0] * lv_, dp_pf[1] * lv_);
        pdz[1] = pack_bf16x2(dp_pf[2] * lv_, nout == 6 ? dout[sc_ * 6 + 3] * lv_ : 0.0f);
        if (nout == 6) pdz[2] = pack_bf16x2(dout[sc_ * 6 + 4] * lv_, dout[sc_ * 6 + 5] * lv_);
      }
      silu_pass<PREC>(zb[0], h, ph, dnext);       // H_4 = silu(Z_3), and silu'(Z_3)
      zstash_load_layer(ztile_, lane, 1, zb[0]);  // z1: used two layers from now
    };
    if constexpr (STASHED) {
      const int64_t tile0 = (int64_t)blockIdx.x * 4 + wid;
      const char* zt0 = zstash + (size_t)(tile0 < ntiles ? tile0 : ntiles - 1) * ZSTASH_TILE;
      zstash_load_layer(zt0, lane, 3, zb[0]);
      zstash_load_layer(zt0, lane, 2, zb[1]);
      head_of_round(0);
    }
    for (int64_t rd = 0; rd < rounds; rd++) {
      asm volatile("" : "+v"(SL.rowbase), "+v"(SL.swz8));  // opaque per round: no hoisting of the ~90 store addresses
      const int64_t tile = rd * nchain + (int64_t)blockIdx.x * 4 + wid;
      const bool active = tile < ntiles;
      const int64_t s = tile * 32 + col;
      const bool live = active && s < n;
      const int64_t sc = live ? s : n - 1;
      // STASHED: the pre-activations parked by the training forward (k_mlp_fwd_stash) arrive one layer ahead of their
      // use through two 17-register buffers, each fetch issued two uses before its SiLU pass (z3 and z2 of a round during
      // the previous round's last layers), so no HBM latency is exposed and the four layers never sit in registers together.  Not STASHED: all four are recomputed here.
      Z33h z[STASHED ? 1 : 4];
      f32x16 dh[3];
      float x[9];
      if constexpr (!STASHED) load_rot9(R, sc, x);  // STASHED: fetched behind layer 1, stored with the layer-0 image
      const int64_t tt = STASHED ? tt_pf : clamp_t(t[sc * t_stride], T);
      const float lv = live ? 1.0f : 0.0f;  // dead columns: dZ_4 = 0, hence every dZ_l = 0 and no contribution to any dW sum
      const char* ztile = STASHED ? zstash + (size_t)(active ? tile : ntiles - 1) * ZSTASH_TILE : nullptr;
      // this lane's 48 input slots of the layer-0 image (bf16 bits as the image wants them), fetched now, stored five layers
      // later: six 16-byte loads instead of 56 four-byte gathers waited for on the spot
      uint4 hq[6];
#pragma unroll
      for (int i = 0; i < 6; i++) hq[i] = h0_tab[(size_t)tt * 12 + 6 * h + i];
      if constexpr (STASHED) {
      } else {
        f32x16 a3[3];
        Tile<PREC> cur;
        layer0_chain<PREC, 0>(lds, beff_tab + (size_t)tt * 96, x, a3, lane);
        keep_h(a3, z[0]);
#pragma unroll
        for (int l = 1; l < 4; l++) {
          activate_zh<PREC>(z[l - 1], cur, h);
          hidden_layer<PREC, 3>(lds + (size_t)frag_hidden<PREC, VAR>(l) * FB, cur, a3, lane);
          keep_h(a3, z[l]);
        }
      }
      // ---- backward, software-pipelined one layer ahead so that the chain waves work while the dW waves
      //      consume the images:   [write images of layer l] B1 [dH_l, dZ_{l-1}, silu pass for layer l-1] B2
      if constexpr (!STASHED) {
#pragma unroll
        for (int r = 0; r < 17; r++) pdz[r] = 0u;
        if (h == 0) {  // dZ_4 = dout in K slots / image columns 0..5 of the lower half (head_of_row)
          const float* dp = dout + sc * nout;
          pdz[0] = pack_bf16x2(dp[0] * lv, dp[1] * lv);
          pdz[1] = pack_bf16x2(dp[2] * lv, nout == 6 ? dp[3] * lv : 0.0f);
          if (nout == 6) pdz[2] = pack_bf16x2(dp[4] * lv, dp[5] * lv);
        }
        silu_pass<PREC, true>(z[3], h, ph, dnext, lv);
      }
#pragma unroll
      for (int l = 4; l >= 0; l--) {
        // ---- images of layer l (8-byte stores of ready-made operand bits)
#pragma unroll
        for (int c8 = 0; c8 < 8; c8++) fimg_store_pk(my_img, SL, 2 * c8 + h, pdz[2 * c8], pdz[2 * c8 + 1]);
        fimg_store_pk(my_img, SL, 16 + h, h ? 0u : pdz[16], 0u);
        if (l > 0) {
#pragma unroll
          for (int c8 = 0; c8 < 8; c8++) fimg_store_pk(my_img, SL, 24 + 2 * c8 + h, ph[2 * c8], ph[2 * c8 + 1]);
          fimg_store_pk(my_img, SL, 24 + 16 + h, ph[16], 0u);
        } else {  // H_0 = the network input: [0..8] R, [9] one, [10..65] emb(t), zeros; the two lanes of a column split the row
          const uint32_t hd[24] = {hq[0].x, hq[0].y, hq[0].z, hq[0].w, hq[1].x, hq[1].y, hq[1].z, hq[1].w, hq[2].x, hq[2].y, hq[2].z, hq[2].w,
                                   hq[3].x, hq[3].y, hq[3].z, hq[3].w, hq[4].x, hq[4].y, hq[4].z, hq[4].w, hq[5].x, hq[5].y, hq[5].z, hq[5].w};
#pragma unroll
          for (int c4 = 0; c4 < 12; c4++) {
            const int ch0 = 24 + 12 * h + c4;  // chunk of the 4 consecutive input slots 48 h + 4 c4 ..
            uint32_t lo = hd[2 * c4], hi = hd[2 * c4 + 1];
            if (c4 < 3) {  // the lower half's first three chunks carry the rotation entries and the constant one
              const uint32_t plo = c4 == 0 ? pack_bf16x2(x[0], x[1]) : (c4 == 1 ? pack_bf16x2(x[4], x[5]) : pack_bf16x2(x[8], 1.0f));
              const uint32_t phi = c4 == 0 ? pack_bf16x2(x[2], x[3]) : (c4 == 1 ? pack_bf16x2(x[6], x[7]) : hi);
              lo = h ? lo : plo;
              hi = h ? hi : phi;
            }
            fimg_store_pk(my_img, SL, ch0, live ? lo : 0u, live ? hi : 0u);
          }
        }
        __syncthreads();  // B1: images of layer l complete -- the dW waves consume them while this wave goes on
        if (l > 0) {
          if (l == 4) dh_layer_pk<PREC, 4>(wt_lds, pdz, dh, lane);
          if (l == 3) dh_layer_pk<PREC, 3>(wt_lds, pdz, dh, lane);
          if (l == 2) dh_layer_pk<PREC, 2>(wt_lds, pdz, dh, lane);
          if (l == 1) dh_layer_pk<PREC, 1>(wt_lds, pdz, dh, lane);
#pragma unroll
          for (int r = 0; r < 16; r++) {
            const float g0 = (r < 8 ? dh[0][2 * r] : dh[1][2 * r - 16]) * dnext[2 * r];
            const float g1 = (r < 8 ? dh[0][2 * r + 1] : dh[1][2 * r - 15]) * dnext[2 * r + 1];
            pdz[r] = pack_bf16x2(g0, g1);
          }
          pdz[16] = pack_bf16x2(h ? 0.0f : dh[2][0] * dnext[32], 0.0f);  // upper half of tile 2 / reg 0 = the constant-one row
          if constexpr (STASHED) {
            // every fetch is issued two uses (~two layers, more than an HBM round trip) ahead:  l = 4 uses z2 (buffer 1)
            // and fetches z0 into it;  l = 3 uses z1 (buffer 0, fetched at the top of the round) and fetches the NEXT
            // round's z3;  l = 2 uses z0 and fetches the next round's z2
            if (l > 1) silu_pass<PREC>(zb[l & 1 ? 0 : 1], h, ph, dnext);
            const int64_t ntile = (rd + 1) * nchain + (int64_t)blockIdx.x * 4 + wid;
            const char* nz = zstash + (size_t)(ntile < ntiles ? ntile : ntiles - 1) * ZSTASH_TILE;
            if (l == 4) zstash_load_layer(ztile, lane, 0, zb[1]);
            if (l == 3) zstash_load_layer(nz, lane, 3, zb[0]);
            if (l == 2) zstash_load_layer(nz, lane, 2, zb[1]);
            if (l == 2) {  // next round's timestep and dout
              bool nl;
              const int64_t sn = sample_of(rd + 1 < rounds ? rd + 1 : rd, &nl);
              tt_pf = clamp_t(t[sn * t_stride], T);
              if (h == 0) { dp_pf[0] = dout[sn * nout]; dp_pf[1] = dout[sn * nout + 1]; dp_pf[2] = dout[sn * nout + 2]; }
            }
            if (l == 1) load_rot9(R, sc, x);
          } else {
            if (l > 1) silu_pass<PREC, true>(z[l - 2], h, ph, dnext, lv);     // H_{l-1} = silu(Z_{l-2}), silu'(Z_{l-2})
          }
        } else if constexpr (STASHED) {
          head_of_round(rd + 1);  // the next round's head, while the dW waves take this round's layer-0 products
        }
        __syncthreads();  // B2: the dW waves are done with the images
      }
    }
  } else {
    // ================================ dW waves =================================
    // one instantiation per dW wave: with the wave's index a compile-time constant its pairs, their layers and every LDS
    // offset fold into the instruction stream
    switch (wid - 4) {
      case 0: dw_role<PREC, 0>(fimg_all, rounds, slabs, lane, nout); break;
      case 1: dw_role<PREC, 1>(fimg_all, rounds, slabs, lane, nout); break;
      case 2: dw_role<PREC, 2>(fimg_all, rounds, slabs, lane, nout); break;
      default: dw_role<PREC, 3>(fimg_all, rounds, slabs, lane, nout); break;
    }
  }
}

// k_bwd_fused + the slab reduction for the whole batch; the images and tables are where launch_prep left them.
inline int launch_fused_bwd(hipStream_t s, const char* img, const char* wt, const float* beff, const float* emb, const uint4* h0,
                            const float* R, const int64_t* t, int64_t t_stride, const float* dout, float* slabs, int64_t n,
                            const char* zstash, int nout, float* dparams, const float* gscale, int T) {
  constexpr int PREC = SO3X_PREC_BF16;
  constexpr int FUSED_LDS = image_bytes<PREC, GATHER>() + wt_bytes<PREC>() + 4 * FIMG_BYTES;
  static PerDevice attr_f0, attr_f1;
  if (int rc2 = ensure_dyn_lds(attr_f0, reinterpret_cast<const void*>(&k_bwd_fused<PREC, false>), FUSED_LDS)) return rc2;
  if (int rc2 = ensure_dyn_lds(attr_f1, reinterpret_cast<const void*>(&k_bwd_fused<PREC, true>), FUSED_LDS)) return rc2;
  const int64_t nt = (n + 31) / 32;
  const int gf = (int)((nt + 3) / 4 < DW_BLOCKS ? (nt + 3) / 4 : DW_BLOCKS);
  if (zstash)
    hipLaunchKernelGGL((k_bwd_fused<PREC, true>), dim3(gf), dim3(512), FUSED_LDS, s, (const void*)img, (const void*)wt, beff,
                       emb, R, t, t_stride, dout, slabs, n, zstash, h0, nout, T);
  else
    hipLaunchKernelGGL((k_bwd_fused<PREC, false>), dim3(gf), dim3(512), FUSED_LDS, s, (const void*)img, (const void*)wt, beff,
                       emb, R, t, t_stride, dout, slabs, n, zstash, h0, nout, T);
  if (!dparams) return check_launch();  // partial slabs only: launch_slab_reduce follows (so3x_train_bwd_reduce)
  hipLaunchKernelGGL(k_bwd_reduce, dim3(red_blocks(nparams(nout))), dim3(RED_THREADS), 0, s, (const float*)slabs, gf, dparams, 0, nparams(nout),
                     gscale);
  return check_launch();
}

// the fixed-order sum of the partial slabs k_bwd_fused left for n samples (same grid rule as launch_fused_bwd)
inline int launch_slab_reduce(hipStream_t s, const float* slabs, int64_t n, int nout, float* dparams, const float* gscale,
                              const float* status = nullptr) {
  const int64_t nt = (n + 31) / 32;
  const int gf = (int)((nt + 3) / 4 < DW_BLOCKS ? (nt + 3) / 4 : DW_BLOCKS);
  hipLaunchKernelGGL(k_bwd_reduce, dim3(red_blocks(nparams(nout))), dim3(RED_THREADS), 0, s, slabs, gf, dparams, 0, nparams(nout), gscale, status);
  return check_launch();
}

// workspace layout: [weight image | (t_table: beff, emb tables)] [transposed image] [slabs] [stash]
struct BwdLayout { size_t wt, slabs, stash, end; };
template <int PREC> BwdLayout bwd_layout(int64_t n, int t_table) {
  const int var = t_table > 0 ? GATHER : FULL;
  const size_t head = t_table > 0 ? tables_end(PREC, var, t_table) : image_bytes_rt(PREC, var);
  BwdLayout L;
  L.wt = (head + 255) & ~(size_t)255;
  L.slabs = (L.wt + (size_t)wt_nfrags<PREC>() * frag_bytes<PREC>() + 255) & ~(size_t)255;
  L.stash = L.slabs + (size_t)DW_BLOCKS * NPARAMS_MAX * sizeof(float);
  const int64_t nc = n < CHUNK ? (n < 32 ? 32 : n) : CHUNK;
  L.end = L.stash + (size_t)STASH_ROWS * (size_t)nc * sizeof(typename Stash<PREC>::T) + 256;
  return L;
}

template <int PREC, int VAR>
int launch_bwd(hipStream_t s, const float* params, const float* R, const int64_t* t, int64_t t_stride, const float* dout,
               float* dparams, int64_t n, int nout, int t_table, char* ws, const char* zstash = nullptr) {
  using ST = typename Stash<PREC>::T;
  constexpr int DW_LDS = PREC == SO3X_PREC_F32 ? 2 * STASH_ROWS * LROW * (int)sizeof(float) : STASH_ROWS * LROW16 * 2;  // fp32: double-buffered
  constexpr int STAGE_LDS = stage_lds_bytes<PREC, VAR>();
  const BwdLayout L = bwd_layout<PREC>(n, t_table);
  static PerDevice attr_stage, attr_dw;
  if (int rc = ensure_dyn_lds(attr_stage, reinterpret_cast<const void*>(&k_bwd_stage<PREC, VAR>), STAGE_LDS)) return rc;
  if (int rc = ensure_dyn_lds(attr_dw, PREC == SO3X_PREC_F32 ? reinterpret_cast<const void*>(&k_bwd_dw)
                                                             : reinterpret_cast<const void*>(&k_bwd_dw_bf16), DW_LDS)) return rc;
  // one prep launch: forward image (not needed when the pre-activations come from a stash), transposed image, tables
  int rc = launch_prep(s, params, PREC, VAR, t_table, ws, nout, (void*)(ws + L.wt), zstash == nullptr);
  if (rc) return rc;
  const float* beff = VAR == GATHER ? reinterpret_cast<const float*>(ws + beff_offset(PREC, VAR)) : nullptr;
  const float* emb = VAR == GATHER ? reinterpret_cast<const float*>(ws + emb_offset(PREC, VAR, t_table)) : nullptr;
  const uint4* h0 = VAR == GATHER ? reinterpret_cast<const uint4*>(ws + h0_offset(PREC, VAR, t_table)) : nullptr;
  float* slabs = reinterpret_cast<float*>(ws + L.slabs);
  ST* stash = reinterpret_cast<ST*>(ws + L.stash);
  if constexpr (PREC == SO3X_PREC_BF16 && VAR == GATHER)  // fused path: no dZ/H stash, one launch for the whole batch
    return launch_fused_bwd(s, ws, ws + L.wt, beff, emb, h0, R, t, t_stride, dout, slabs, n, zstash, nout, dparams, nullptr, t_table);
  for (int64_t c0 = 0; c0 < n; c0 += CHUNK) {
    const int64_t nc = (n - c0) < CHUNK ? (n - c0) : CHUNK;
    const int64_t ntiles = (nc + 31) / 32;
    const int g1 = (int)((ntiles + 3) / 4 < 256 ? (ntiles + 3) / 4 : 256);
    hipLaunchKernelGGL((k_bwd_stage<PREC, VAR>), dim3(g1), dim3(256), STAGE_LDS, s, (const void*)ws,
                       (const void*)(ws + L.wt), beff, emb, R + c0 * 9, t + (t_stride ? c0 : 0), t_stride, dout + c0 * nout,
                       host_freqs(), stash, nc, nout, t_table);
    const int g2 = (int)(ntiles < DW_BLOCKS ? ntiles : DW_BLOCKS);
    if constexpr (PREC == SO3X_PREC_F32)
      hipLaunchKernelGGL(k_bwd_dw, dim3(g2), dim3(512), DW_LDS, s, (const float*)stash, nc, slabs, nout);
    else
      hipLaunchKernelGGL(k_bwd_dw_bf16, dim3(g2), dim3(512), DW_LDS, s, (const __bf16*)stash, nc, slabs, nout);
    hipLaunchKernelGGL(k_bwd_reduce, dim3(red_blocks(nparams(nout))), dim3(RED_THREADS), 0, s, (const float*)slabs, g2, dparams,
                       c0 > 0 ? 1 : 0, nparams(nout));
  }
  return check_launch();
}

}  // namespace

extern "C" {

size_t so3x_mlp_workspace_bytes(int64_t n, int precision, int t_table) {
  if (precision == SO3X_PREC_F32) return bwd_layout<SO3X_PREC_F32>(n, t_table).end;
  return bwd_layout<SO3X_PREC_BF16>(n, t_table).end;
}

size_t so3x_mlp_stash_bytes(int64_t n) { return (size_t)((n > 0 ? n : 0) + 31) / 32 * ZSTASH_TILE; }

int so3x_mlp_fwd_stash(so3x_stream_t s, const float* params, const float* R, const int64_t* t, int64_t t_stride, float* out,
                       void* zstash, int64_t n, int n_out, int precision, int t_table, void* workspace, size_t workspace_bytes) {
  if (n < 0 || (n && (!params || !R || !t || !out || !zstash)) || (t_stride != 0 && t_stride != 1) || t_table < 0 ||
      (n_out != 3 && n_out != 6))
    return SO3X_ERR_INVALID_ARG;
  if (precision != SO3X_PREC_BF16 || t_table <= 0) return SO3X_ERR_UNSUPPORTED;  // the stash is the fused backward's
  if (!workspace || workspace_bytes < tables_end(precision, GATHER, t_table)) return SO3X_ERR_WORKSPACE;
  if (n == 0) return SO3X_OK;
  constexpr int PREC = SO3X_PREC_BF16, IMG = image_bytes<PREC, GATHER_T>();
  char* ws = (char*)workspace;
  int rc = launch_prep((hipStream_t)s, params, PREC, GATHER_T, t_table, ws, n_out);
  if (rc) return rc;
  static PerDevice attr;
  if ((rc = ensure_dyn_lds(attr, reinterpret_cast<const void*>(&k_mlp_fwd_stash<PREC, false>), IMG))) return rc;
  const int64_t ntiles = (n + 31) / 32, want = (ntiles + 7) / 8;
  hipLaunchKernelGGL((k_mlp_fwd_stash<PREC, false>), dim3((int)(want < 512 ? want : 512)), dim3(kFwdStashThreads), IMG, (hipStream_t)s, (const void*)ws,
                     reinterpret_cast<const float*>(ws + beff_offset(PREC, GATHER)), R, t, t_stride, out, (char*)zstash, n, n_out, LossArgs{}, t_table);
  return check_launch();
}

int so3x_mlp_bwd(so3x_stream_t s, const float* params, const float* R, const int64_t* t, int64_t t_stride,
                 const float* dout, float* dparams, int64_t n, int n_out, int precision, int t_table, const void* zstash,
                 void* workspace, size_t workspace_bytes) {
  if (n < 0 || (n && (!params || !R || !t || !dout)) || !dparams || (t_stride != 0 && t_stride != 1) || t_table < 0 ||
      (zstash && (precision != SO3X_PREC_BF16 || t_table <= 0)) || (n_out != 3 && n_out != 6))
    return SO3X_ERR_INVALID_ARG;
  if (precision != SO3X_PREC_F32 && precision != SO3X_PREC_BF16) return SO3X_ERR_UNSUPPORTED;
  if (!workspace || workspace_bytes < so3x_mlp_workspace_bytes(n, precision, t_table)) return SO3X_ERR_WORKSPACE;
  if (n == 0) {
    hipError_t e = hipMemsetAsync(dparams, 0, nparams(n_out) * sizeof(float), (hipStream_t)s);
    return e == hipSuccess ? SO3X_OK : (int)e;
  }
  char* ws = (char*)workspace;
  hipStream_t st = (hipStream_t)s;
  if (precision == SO3X_PREC_F32)
    return t_table > 0 ? launch_bwd<SO3X_PREC_F32, GATHER>(st, params, R, t, t_stride, dout, dparams, n, n_out, t_table, ws)
                       : launch_bwd<SO3X_PREC_F32, FULL>(st, params, R, t, t_stride, dout, dparams, n, n_out, 0, ws);
  return t_table > 0 ? launch_bwd<SO3X_PREC_BF16, GATHER>(st, params, R, t, t_stride, dout, dparams, n, n_out, t_table, ws, (const char*)zstash)
                     : launch_bwd<SO3X_PREC_BF16, FULL>(st, params, R, t, t_stride, dout, dparams, n, n_out, 0, ws);
}

// ---- one training step of SO3Diffusion(RotPredict(out_type="skewvec")) under loss_type="skewvec", bf16 MLP operands ----
size_t so3x_train_workspace_bytes(int64_t n, int T) { return train_layout(n, T > 0 ? T : 0).end; }

// The step in stages.  A captured data-parallel step pipelines them: the noising of batch k+1 (a function of the data and the
// Philox counter only) runs on a second stream beside [slab reduction -> gradient all-reduce -> Adam] of batch k.
int so3x_train_noise(so3x_stream_t s, const float* sched, int T, const float* trap_q, const uint16_t* guide_q, const float* x0,
                     const int64_t* t, int64_t* t_used, int quirk_col0, const float* axes, const float* unif, uint64_t seed,
                     uint64_t rng_offset, const int64_t* rng_counter, int64_t index_base, int64_t n, float* x_t, void* workspace,
                     size_t workspace_bytes) {
  if (n <= 0 || T <= 0 || !sched || !trap_q || !x0 || !t_used || !x_t || ((axes == nullptr) != (unif == nullptr))) return SO3X_ERR_INVALID_ARG;
  const TrainLayout L = train_layout(n, T);
  if (!workspace || workspace_bytes < L.end) return SO3X_ERR_WORKSPACE;
  // the noising kernel (eight waves per SIMD: latency-bound gathers) leaves x_t, the timesteps used and the regression target
  return launch_q_sample_target((hipStream_t)s, sched, T, trap_q, guide_q, x0, t, t_used, quirk_col0, nullptr, axes, unif, seed, rng_offset,
                                rng_counter, index_base, x_t, reinterpret_cast<float*>((char*)workspace + L.target), nullptr, n);
}

int so3x_train_net(so3x_stream_t s, const float* params, int T, const float* x_t, const int64_t* t_used, int64_t n, float* dout,
                   void* zstash, float* loss, float* out, int64_t* rng_counter, void* workspace, size_t workspace_bytes) {
  if (n <= 0 || T <= 0 || !params || !x_t || !t_used || !dout || !zstash || !loss) return SO3X_ERR_INVALID_ARG;
  const TrainLayout L = train_layout(n, T);
  if (!workspace || workspace_bytes < L.end) return SO3X_ERR_WORKSPACE;
  constexpr int PREC = SO3X_PREC_BF16, IMG = image_bytes<PREC, GATHER_T>();
  char* ws = (char*)workspace;
  hipStream_t st = (hipStream_t)s;
  // one prep launch for the whole step: forward image, transposed image of the backward, per-timestep tables
  int rc = launch_prep(st, params, PREC, GATHER_T, T, ws, 3, (void*)(ws + L.wt), true, reinterpret_cast<unsigned*>(ws + L.ticket));
  if (rc) return rc;
  LossArgs la;
  la.dout = dout; la.loss = loss;
  la.partial = reinterpret_cast<double*>(ws + L.partial);
  la.status = reinterpret_cast<float*>(ws + L.ticket + 64);
  la.ticket = reinterpret_cast<unsigned*>(ws + L.ticket);
  la.rng_counter = rng_counter;
  la.dscale = (float)(2.0 / (3.0 * (double)n));
  la.inv_count = 1.0 / (3.0 * (double)n);
  la.target = reinterpret_cast<float*>(ws + L.target);
  const float* beff = reinterpret_cast<const float*>(ws + beff_offset(PREC, GATHER));
  static PerDevice attr;
  if ((rc = ensure_dyn_lds(attr, reinterpret_cast<const void*>(&k_mlp_fwd_stash<PREC, true>), IMG))) return rc;
  const int64_t ntiles = (n + 31) / 32, want = (ntiles + 7) / 8;
  hipLaunchKernelGGL((k_mlp_fwd_stash<PREC, true>), dim3((int)(want < 512 ? want : 512)), dim3(kFwdStashThreads), IMG, st, (const void*)ws,
                     beff, x_t, t_used, (int64_t)1, out, (char*)zstash, n, 3, la, T);
  return check_launch();
}

int so3x_train_fwd(so3x_stream_t s, const float* params, const float* sched, int T, const float* trap_q, const uint16_t* guide_q,
                   const float* x0, const int64_t* t, int64_t* t_used, int quirk_col0, const float* axes, const float* unif, uint64_t seed,
                   uint64_t rng_offset, int64_t* rng_counter, int64_t index_base, int64_t n, float* x_t, float* dout, void* zstash,
                   float* loss, float* out, void* workspace, size_t workspace_bytes) {
  if (n <= 0 || T <= 0 || !params || !sched || !trap_q || !x0 || !t_used || !x_t || !dout || !zstash || !loss ||
      ((axes == nullptr) != (unif == nullptr)))
    return SO3X_ERR_INVALID_ARG;
  int64_t* counter = (axes == nullptr || t == nullptr) ? rng_counter : nullptr;  // advanced only by a call that drew from it
  if (int rc = so3x_train_noise(s, sched, T, trap_q, guide_q, x0, t, t_used, quirk_col0, axes, unif, seed, rng_offset, rng_counter, index_base, n,
                                x_t, workspace, workspace_bytes))
    return rc;
  return so3x_train_net(s, params, T, x_t, t_used, n, dout, zstash, loss, out, counter, workspace, workspace_bytes);
}

int so3x_train_bwd_partial(so3x_stream_t s, const float* x_t, const int64_t* t, const float* dout, const void* zstash, int64_t n, int T,
                           void* workspace, size_t workspace_bytes) {
  if (n <= 0 || T <= 0 || !x_t || !t || !dout || !zstash) return SO3X_ERR_INVALID_ARG;
  const TrainLayout L = train_layout(n, T);
  if (!workspace || workspace_bytes < L.end) return SO3X_ERR_WORKSPACE;
  constexpr int PREC = SO3X_PREC_BF16;
  const char* ws = (const char*)workspace;
  return launch_fused_bwd((hipStream_t)s, ws, ws + L.wt, reinterpret_cast<const float*>(ws + beff_offset(PREC, GATHER)),
                          reinterpret_cast<const float*>(ws + emb_offset(PREC, GATHER, T)),
                          reinterpret_cast<const uint4*>(ws + h0_offset(PREC, GATHER, T)), x_t, t, 1, dout,
                          reinterpret_cast<float*>(const_cast<char*>(ws) + L.slabs), n, (const char*)zstash, 3, nullptr, nullptr, T);
}

int so3x_train_bwd_reduce(so3x_stream_t s, int64_t n, int T, const float* gscale, float* grad, const void* workspace, size_t workspace_bytes) {
  if (n <= 0 || T <= 0 || !grad) return SO3X_ERR_INVALID_ARG;
  const TrainLayout L = train_layout(n, T);
  if (!workspace || workspace_bytes < L.end) return SO3X_ERR_WORKSPACE;
  return launch_slab_reduce((hipStream_t)s, reinterpret_cast<const float*>((const char*)workspace + L.slabs), n, 3, grad, gscale,
                            reinterpret_cast<const float*>((const char*)workspace + L.ticket + 64));
}

int so3x_train_bwd_reduce_adam(so3x_stream_t s, int64_t n, int T, const float* gscale, float* grad, const void* workspace,
                               size_t workspace_bytes, float* params, float* exp_avg, float* exp_avg_sq, float* step, float lr, float beta1,
                               float beta2, float eps, float weight_decay, float grad_scale) {
  if (n <= 0 || T <= 0 || !grad || !params || !exp_avg || !exp_avg_sq || !step) return SO3X_ERR_INVALID_ARG;
  const TrainLayout L = train_layout(n, T);
  if (!workspace || workspace_bytes < L.end) return SO3X_ERR_WORKSPACE;
  const int64_t nt = (n + 31) / 32;
  const int gf = (int)((nt + 3) / 4 < DW_BLOCKS ? (nt + 3) / 4 : DW_BLOCKS);
  AdamArgs ad{params, exp_avg, exp_avg_sq, step, reinterpret_cast<unsigned*>(step + 1), lr, beta1, beta2, eps, weight_decay, grad_scale};
  hipLaunchKernelGGL(k_bwd_reduce_adam, dim3(red_blocks(nparams(3))), dim3(RED_THREADS), 0, (hipStream_t)s,
                     reinterpret_cast<const float*>((const char*)workspace + L.slabs), gf, grad, nparams(3), gscale, ad,
                     reinterpret_cast<const float*>((const char*)workspace + L.ticket + 64));
  return check_launch();
}

int so3x_train_bwd(so3x_stream_t s, const float* x_t, const int64_t* t, const float* dout, const void* zstash, int64_t n, int T,
                   const float* gscale, float* grad, void* workspace, size_t workspace_bytes) {
  if (!grad) return SO3X_ERR_INVALID_ARG;
  if (int rc = so3x_train_bwd_partial(s, x_t, t, dout, zstash, n, T, workspace, workspace_bytes)) return rc;
  return so3x_train_bwd_reduce(s, n, T, gscale, grad, workspace, workspace_bytes);
}

}  // extern "C"
