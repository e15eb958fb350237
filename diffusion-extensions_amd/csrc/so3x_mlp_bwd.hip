// so3x_mlp_bwd.hip -- backward of the RotPredict score network (autograd of
// so3_train.py:39-49; only the 17,358 parameters need gradients).
//
// Three kernels per chunk of <= 65,536 samples (activations are recomputed, never saved
// by the forward op):
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
// The stash is fp32 in both precisions; K2/K3 always run exact-fp32 MFMA.
#include "so3x_common.hpp"
#include "so3x_mlp.hpp"

using namespace so3x;
using namespace so3x::mlp;

namespace {

constexpr int HROWS = 66;                       // 65 inputs + the constant one (bias carrier)
constexpr int H_BASE(int l) { return HROWS * l; }
constexpr int DZ_BASE(int l) { return 5 * HROWS + D * l; }
constexpr int STASH_ROWS = 5 * HROWS + 4 * D + 3;  // 593
constexpr int CHUNK = 1 << 19;  // samples per stash chunk: 593 rows x 2^19 x 4 B = 1.24 GB of workspace (288 GB HBM)
constexpr int DW_BLOCKS = 256;
constexpr int NPAIRS = 39;                      // 4 layers x 3x3 tiles + last layer 1x3

// ---- transposed-weight image (A operand of dH = W^T dZ), global/L2-resident -----------
// fragment order: layers 1..3: [l-1][To(in-feature tile) 3][ks over out-features KH], then layer 4: [To 3][K4]
template <int PREC> __host__ __device__ constexpr int k4() { return PREC == SO3X_PREC_F32 ? 3 : 1; }
template <int PREC> __host__ __device__ constexpr int wt_frag(int l, int to, int ks) {
  return l < 4 ? ((l - 1) * 3 + to) * ks_hidden<PREC>() + ks : 9 * ks_hidden<PREC>() + to * k4<PREC>() + ks;
}
template <int PREC> __host__ __device__ constexpr int wt_nfrags() { return 9 * ks_hidden<PREC>() + 3 * k4<PREC>(); }

template <int PREC> __global__ void __launch_bounds__(256) k_prep_wt(const float* __restrict__ params, void* __restrict__ img) {
  constexpr int EPL = PREC == SO3X_PREC_F32 ? 1 : 8;
  constexpr int KH = ks_hidden<PREC>();
  const int total = wt_nfrags<PREC>() * 64 * EPL;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
    const int j = e % EPL, lane = (e / EPL) % 64, frag = e / (EPL * 64);
    const int i = lane & 31, h = lane >> 5;
    int l, to, ks;
    if (frag < 9 * KH) { l = 1 + frag / (3 * KH); to = (frag % (3 * KH)) / KH; ks = frag % KH; }
    else { const int f = frag - 9 * KH; l = 4; to = f / k4<PREC>(); ks = f % k4<PREC>(); }
    const int in = 32 * to + i;                       // row of W^T = input feature of layer l
    const int out = hidden_feature<PREC>(ks, h, j);   // k index = output feature of layer l
    float v = 0.0f;
    if (in < D && out < dout_of(l)) v = params[l * LAYER_STRIDE + out * D + in];
    if (PREC == SO3X_PREC_F32) reinterpret_cast<float*>(img)[e] = v;
    else reinterpret_cast<__bf16*>(img)[e] = (__bf16)v;
  }
}

struct Z33 { float v[33]; };  // [0..15] tile 0, [16..31] tile 1, [32] = tile 2 reg 0 (feature 64 in the lower half)

// silu and its derivative from the pre-activation
template <int PREC> __device__ __forceinline__ void silu_grad(float z, float* hval, float* dval) {
  float sg;
  if (PREC == SO3X_PREC_F32) sg = 1.0f / (1.0f + expf(-z));
  else sg = __builtin_amdgcn_rcpf(1.0f + __expf(-z));
  const float hv = z * sg;
  *hval = hv;
  *dval = sg + hv * (1.0f - sg);  // sigma (1 + z (1 - sigma))
}

// dH = W^T dZ for one layer; dz in the 33-register layout of Z33 (layer-4: only v[0..2] of the lower half)
template <int PREC, int L>
__device__ __forceinline__ void dh_layer(const void* __restrict__ wt, const Z33& dz, f32x16 (&dh)[3], int lane) {
  if constexpr (PREC == SO3X_PREC_F32) {
    const float* w = reinterpret_cast<const float*>(wt);
    constexpr int KS = L < 4 ? 33 : 3;
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
  (stash + (int64_t)row0 * nc)[off] = (T)v;
}

// stash row of feature-layout register q (Z33 order) for lane half h: rows 0..63 for q < 32; q == 32: feature 64 (h = 0)
__device__ __forceinline__ int z33_row0(int q) { return q < 32 ? 32 * (q >> 4) + row_of(q & 15, 0) : 64; }

// ---------------------------------------------------------------------------------------
// K1.  LDS holds the forward image and the transposed image.  In bf16 both fit (65 + 48 KB)
// and the waves run free; in fp32 (110 + 78 KB > 160 KB) the block swaps the two images
// around a barrier once per tile round (4 tiles per block per round).
// Pre-activations are kept as 33 live registers per layer (tiles 0,1 + feature 64).
// ---------------------------------------------------------------------------------------
template <int PREC> __host__ __device__ constexpr int wt_bytes() { return wt_nfrags<PREC>() * frag_bytes<PREC>(); }
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
            int64_t nc /*samples in this chunk*/) {
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
  const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
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
    float d0 = 0.f, d1 = 0.f, d2 = 0.f;
    if (active) {
      float x[9];
#pragma unroll
      for (int j = 0; j < 9; j++) x[j] = R[sc * 9 + j];
      const int64_t tt = t[sc * t_stride];
      d0 = dout[sc * 3]; d1 = dout[sc * 3 + 1]; d2 = dout[sc * 3 + 2];
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
        dz.v[0] = d0; dz.v[1] = d1; dz.v[2] = d2;
        if (live) {
          st_row(stash, nc, DZ_BASE(4) + 0, off1, d0);  // h == 0 here: off1 == s
          st_row(stash, nc, DZ_BASE(4) + 1, off1, d1);
          st_row(stash, nc, DZ_BASE(4) + 2, off1, d2);
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

__device__ __forceinline__ void write_slab_impl(float* __restrict__ slabs, const f32x16 (&acc)[5], int wid, int i, int h);
__device__ __forceinline__ void write_slab(float* __restrict__ slabs, const f32x16 (&acc)[5], int wid, int i, int h) {
  write_slab_impl(slabs, acc, wid, i, h);
}

constexpr int LROW = 33;  // padded LDS row (floats): fragment reads of stride-LROW rows are conflict-free

__global__ void __launch_bounds__(512, 1)
k_bwd_dw(const float* __restrict__ stash, int64_t nc, float* __restrict__ slabs) {
  extern __shared__ __attribute__((aligned(16))) float sm[];  // [STASH_ROWS][LROW]
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int i = lane & 31, h = lane >> 5;
  f32x16 acc[5];
#pragma unroll
  for (int k = 0; k < 5; k++) acc[k] = zero16<0>();
  const int64_t nsub = (nc + 31) / 32;
  for (int64_t sub = blockIdx.x; sub < nsub; sub += gridDim.x) {
    const int64_t s0 = sub * 32;
    __syncthreads();
    // stage [593 rows][32 samples]: 8 lanes x 16 B per row, coalesced along the sample axis
    for (int e = threadIdx.x; e < STASH_ROWS * 8; e += blockDim.x) {
      const int row = e >> 3, c4 = (e & 7) * 4;
      const float* src = stash + (int64_t)row * nc + s0 + c4;
      float v[4];
      if (s0 + c4 + 3 < nc && ((reinterpret_cast<uintptr_t>(src) & 15) == 0)) {
        const float4 q = *reinterpret_cast<const float4*>(src);
        v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
      } else {
#pragma unroll
        for (int u = 0; u < 4; u++) v[u] = (s0 + c4 + u < nc) ? src[u] : 0.0f;
      }
#pragma unroll
      for (int u = 0; u < 4; u++) sm[row * LROW + c4 + u] = v[u];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 5; k++) {
      const int p = wid + 8 * k;
      if (p < NPAIRS) {
        int l, to, ti;
        pair_of(p, &l, &to, &ti);
        const int nout = (l < 4 ? D : 3) - 32 * to;   // valid dZ rows in this tile
        const int nin = HROWS - 32 * ti;              // valid H rows in this tile
        const bool va = i < nout, vb = i < nin;
        const float* pa = sm + (DZ_BASE(l) + 32 * to + (va ? i : 0)) * LROW + h;
        const float* pb = sm + (H_BASE(l) + 32 * ti + (vb ? i : 0)) * LROW + h;
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
  }
  write_slab(slabs, acc, wid, i, h);
}

// this block's partial dparams slab (every entry is owned by exactly one lane of one wave)
__device__ __forceinline__ void write_slab_impl(float* __restrict__ slabs, const f32x16 (&acc)[5], int wid, int i, int h) {
  float* slab = slabs + (size_t)blockIdx.x * NPARAMS;
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
        if (out >= (l < 4 ? D : 3) || in_ext >= HROWS) continue;
        int col;  // column of the layer's weight, or -2 = bias
        if (l == 0) col = in_ext < 9 ? in_ext : (in_ext == 9 ? -2 : in_ext - 1);  // [R(9), one, emb(56)] -> cols 0..8, bias, 9..64
        else col = in_ext < D ? in_ext : -2;
        const int base = l * LAYER_STRIDE;
        const int idx = col >= 0 ? base + out * D + col : base + (l < 4 ? D : 3) * D + out;
        slab[idx] = acc[k][r];
      }
    }
  }
}

// bf16 stash variant: rows of 32 bf16 (64 B) staged at an 80-B pitch (16-B aligned, ds_read_b128 conflict-free),
// two v_mfma_f32_32x32x16_bf16 k-steps per 32-sample slice (k = sample index), fp32 accumulation.
constexpr int LROW16 = 40;  // bf16 elements per padded LDS row

__global__ void __launch_bounds__(512, 1)
k_bwd_dw_bf16(const __bf16* __restrict__ stash, int64_t nc, float* __restrict__ slabs) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  __bf16* sb = reinterpret_cast<__bf16*>(sm);  // [STASH_ROWS][LROW16]
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
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
        const int nout = (l < 4 ? D : 3) - 32 * to;
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
  write_slab(slabs, acc, wid, i, h);
}

// 32 parameters x 8 slab groups per block: coalesced 128-B reads, 8-way split of the slab loop, fixed
// summation order (deterministic)
__global__ void __launch_bounds__(256)
k_bwd_reduce(const float* __restrict__ slabs, int nslabs, float* __restrict__ dparams, int accumulate) {
  __shared__ float part[8][33];
  const int p = threadIdx.x & 31, g = threadIdx.x >> 5;
  const int idx = blockIdx.x * 32 + p;
  float s = 0.0f;
  if (idx < NPARAMS)
    for (int b = g; b < nslabs; b += 8) s += slabs[(size_t)b * NPARAMS + idx];
  part[g][p] = s;
  __syncthreads();
  if (g == 0 && idx < NPARAMS) {
    float t = 0.0f;
#pragma unroll
    for (int k = 0; k < 8; k++) t += part[k][p];
    dparams[idx] = accumulate ? dparams[idx] + t : t;
  }
}

// workspace layout: [weight image | (t_table: beff, emb tables)] [transposed image] [slabs] [stash]
struct BwdLayout { size_t wt, slabs, stash, end; };
template <int PREC> BwdLayout bwd_layout(int64_t n, int t_table) {
  const int var = t_table > 0 ? GATHER : FULL;
  const size_t head = t_table > 0 ? tables_end(PREC, var, t_table) : image_bytes_rt(PREC, var);
  BwdLayout L;
  L.wt = (head + 255) & ~(size_t)255;
  L.slabs = (L.wt + (size_t)wt_nfrags<PREC>() * frag_bytes<PREC>() + 255) & ~(size_t)255;
  L.stash = L.slabs + (size_t)DW_BLOCKS * NPARAMS * sizeof(float);
  const int64_t nc = n < CHUNK ? (n < 32 ? 32 : n) : CHUNK;
  L.end = L.stash + (size_t)STASH_ROWS * (size_t)nc * sizeof(typename Stash<PREC>::T) + 256;
  return L;
}

template <int PREC, int VAR>
int launch_bwd(hipStream_t s, const float* params, const float* R, const int64_t* t, int64_t t_stride, const float* dout,
               float* dparams, int64_t n, int t_table, char* ws) {
  using ST = typename Stash<PREC>::T;
  constexpr int DW_LDS = PREC == SO3X_PREC_F32 ? STASH_ROWS * LROW * (int)sizeof(float) : STASH_ROWS * LROW16 * 2;
  constexpr int STAGE_LDS = stage_lds_bytes<PREC, VAR>();
  const BwdLayout L = bwd_layout<PREC>(n, t_table);
  static int attr_set = 0;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_bwd_stage<PREC, VAR>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, STAGE_LDS);
    if (e != hipSuccess) return (int)e;
    e = hipFuncSetAttribute(PREC == SO3X_PREC_F32 ? reinterpret_cast<const void*>(&k_bwd_dw)
                                                  : reinterpret_cast<const void*>(&k_bwd_dw_bf16),
                            hipFuncAttributeMaxDynamicSharedMemorySize, DW_LDS);
    if (e != hipSuccess) return (int)e;
    attr_set = 1;
  }
  int rc = launch_prep(s, params, PREC, VAR, t_table, ws);
  if (rc) return rc;
  hipLaunchKernelGGL((k_prep_wt<PREC>), dim3(32), dim3(256), 0, s, params, (void*)(ws + L.wt));
  const float* beff = VAR == GATHER ? reinterpret_cast<const float*>(ws + beff_offset(PREC, VAR)) : nullptr;
  const float* emb = VAR == GATHER ? reinterpret_cast<const float*>(ws + emb_offset(PREC, VAR, t_table)) : nullptr;
  float* slabs = reinterpret_cast<float*>(ws + L.slabs);
  ST* stash = reinterpret_cast<ST*>(ws + L.stash);
  for (int64_t c0 = 0; c0 < n; c0 += CHUNK) {
    const int64_t nc = (n - c0) < CHUNK ? (n - c0) : CHUNK;
    const int64_t ntiles = (nc + 31) / 32;
    const int g1 = (int)((ntiles + 3) / 4 < 256 ? (ntiles + 3) / 4 : 256);
    hipLaunchKernelGGL((k_bwd_stage<PREC, VAR>), dim3(g1), dim3(256), STAGE_LDS, s, (const void*)ws,
                       (const void*)(ws + L.wt), beff, emb, R + c0 * 9, t + (t_stride ? c0 : 0), t_stride, dout + c0 * 3,
                       host_freqs(), stash, nc);
    const int g2 = (int)(ntiles < DW_BLOCKS ? ntiles : DW_BLOCKS);
    if constexpr (PREC == SO3X_PREC_F32)
      hipLaunchKernelGGL(k_bwd_dw, dim3(g2), dim3(512), DW_LDS, s, (const float*)stash, nc, slabs);
    else
      hipLaunchKernelGGL(k_bwd_dw_bf16, dim3(g2), dim3(512), DW_LDS, s, (const __bf16*)stash, nc, slabs);
    hipLaunchKernelGGL(k_bwd_reduce, dim3((NPARAMS + 31) / 32), dim3(256), 0, s, (const float*)slabs, g2, dparams,
                       c0 > 0 ? 1 : 0);
  }
  return check_launch();
}

}  // namespace

extern "C" {

size_t so3x_mlp_workspace_bytes(int64_t n, int precision, int t_table) {
  if (precision == SO3X_PREC_F32) return bwd_layout<SO3X_PREC_F32>(n, t_table).end;
  return bwd_layout<SO3X_PREC_BF16>(n, t_table).end;
}

int so3x_mlp_bwd(so3x_stream_t s, const float* params, const float* R, const int64_t* t, int64_t t_stride,
                 const float* dout, float* dparams, int64_t n, int precision, int t_table, void* workspace,
                 size_t workspace_bytes) {
  if (n < 0 || (n && (!params || !R || !t || !dout)) || !dparams || (t_stride != 0 && t_stride != 1) || t_table < 0)
    return SO3X_ERR_INVALID_ARG;
  if (precision != SO3X_PREC_F32 && precision != SO3X_PREC_BF16) return SO3X_ERR_UNSUPPORTED;
  if (!workspace || workspace_bytes < so3x_mlp_workspace_bytes(n, precision, t_table)) return SO3X_ERR_WORKSPACE;
  if (n == 0) {
    hipError_t e = hipMemsetAsync(dparams, 0, NPARAMS * sizeof(float), (hipStream_t)s);
    return e == hipSuccess ? SO3X_OK : (int)e;
  }
  char* ws = (char*)workspace;
  hipStream_t st = (hipStream_t)s;
  if (precision == SO3X_PREC_F32)
    return t_table > 0 ? launch_bwd<SO3X_PREC_F32, GATHER>(st, params, R, t, t_stride, dout, dparams, n, t_table, ws)
                       : launch_bwd<SO3X_PREC_F32, FULL>(st, params, R, t, t_stride, dout, dparams, n, 0, ws);
  return t_table > 0 ? launch_bwd<SO3X_PREC_BF16, GATHER>(st, params, R, t, t_stride, dout, dparams, n, t_table, ws)
                     : launch_bwd<SO3X_PREC_BF16, FULL>(st, params, R, t, t_stride, dout, dparams, n, 0, ws);
}

}  // extern "C"
