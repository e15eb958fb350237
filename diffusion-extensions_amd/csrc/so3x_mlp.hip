// so3x_mlp.hip -- standalone score-network ops: weight-image prep, forward, backward.
#include <math.h>
#include "so3x_common.hpp"
#include "so3x_mlp.hpp"

using namespace so3x;
using namespace so3x::mlp;

namespace so3x {
namespace mlp {

const Freqs& host_freqs() {
  static const Freqs fr = [] {
    Freqs f;
    so3x_posemb_freqs(NFREQ, f.f);
    return f;
  }();
  return fr;
}

size_t image_bytes_rt(int precision, int variant) {
  if (precision == SO3X_PREC_F32) return chain_layout(variant) ? image_bytes<SO3X_PREC_F32, CHAIN>() : image_bytes<SO3X_PREC_F32, FULL>();
  // (every chain-layout variant reserves the largest of their images -- GATHER_TD's, with its 4 KB table -- so that the tables
  //  behind the image sit at one offset whichever variant the prep launch wrote)
  return chain_layout(variant) ? image_bytes<SO3X_PREC_BF16, GATHER_TD>() : image_bytes<SO3X_PREC_BF16, FULL>();
}
size_t beff_offset(int precision, int variant) { return (image_bytes_rt(precision, variant) + 255) & ~(size_t)255; }

}  // namespace mlp
}  // namespace so3x

namespace {

// ---- prep (one launch): blocks [0, 32) the permuted/padded weight image, [32, 64) the transposed image of the backward,
//      [64, 64 + T) the per-timestep rows:  beff[t][o] = b_0[o] + sum_e W_0[o][9+e] * emb_e(t)  (o < 65, other rows 0;
//      appendix C.3), emb[t][56] and the bf16 input-slot row h0[t][96] (embedding at slots 10..65).
constexpr int PREP_IMG_BLOCKS = 32, PREP_WT_BLOCKS = 32;
template <int PREC, int VAR>
__global__ void __launch_bounds__(256) k_prep(const float* __restrict__ params, void* __restrict__ img, void* __restrict__ wt, int nout,
                                              Freqs fr, int T, float scale, float offset, float* __restrict__ beff, float* __restrict__ emb_tab,
                                              __bf16* __restrict__ h0_tab, unsigned* __restrict__ zero_word, int t_first, int f16) {
  constexpr int EPL = PREC == SO3X_PREC_F32 ? 1 : 8;  // elements per lane per fragment
  if (zero_word && blockIdx.x == 0 && threadIdx.x == 0) *zero_word = 0u;  // arrival ticket of the launch that follows
  if (blockIdx.x < PREP_IMG_BLOCKS) {
    if (!img) return;
    const int total = n_frags<PREC, VAR>() * 64 * EPL;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < total; e += PREP_IMG_BLOCKS * 256) {
      const int j = e % EPL, lane = (e / EPL) % 64, frag = e / (EPL * 64);
      const float v = image_value<PREC, VAR>(params, frag, lane, j, nout);
      if (PREC == SO3X_PREC_F32) reinterpret_cast<float*>(img)[e] = v;
      else if (f16) reinterpret_cast<_Float16*>(img)[e] = (_Float16)v;
      else reinterpret_cast<__bf16*>(img)[e] = (__bf16)v;
    }
    if (fold_scale<PREC, VAR>() && blockIdx.x == PREP_IMG_BLOCKS - 1) {  // the SiLU table behind the fragments
      char* tab = reinterpret_cast<char*>(img) + (size_t)n_frags<PREC, VAR>() * frag_bytes<PREC>();
      if constexpr (VAR == GATHER_TD) reinterpret_cast<float4*>(tab)[threadIdx.x] = silu_table_entry4(threadIdx.x);
      else reinterpret_cast<float2*>(tab)[threadIdx.x] = silu_table_entry(threadIdx.x);
    }
    return;
  }
  if (blockIdx.x < PREP_IMG_BLOCKS + PREP_WT_BLOCKS) {
    if (!wt) return;
    const int total = wt_nfrags<PREC>() * 64 * EPL;
    for (int e = (blockIdx.x - PREP_IMG_BLOCKS) * 256 + threadIdx.x; e < total; e += PREP_WT_BLOCKS * 256) {
      const int j = e % EPL, lane = (e / EPL) % 64, frag = e / (EPL * 64);
      const float v = wt_value<PREC>(params, frag, lane, j, nout);
      if (PREC == SO3X_PREC_F32) reinterpret_cast<float*>(wt)[e] = v;
      else reinterpret_cast<__bf16*>(wt)[e] = (__bf16)v;
    }
    return;
  }
  __shared__ float emb[NEMB];
  const int t = t_first + blockIdx.x - PREP_IMG_BLOCKS - PREP_WT_BLOCKS;  // per-timestep rows t_first .. (only those the caller will read)
  if (threadIdx.x < NEMB) {
    emb[threadIdx.x] = emb_value((int64_t)t, threadIdx.x, fr);
    if (emb_tab) emb_tab[(size_t)t * NEMB + threadIdx.x] = emb[threadIdx.x];  // [T][56] time-embedding table
  }
  __syncthreads();
  if (h0_tab && threadIdx.x < 96)  // [T][96] bf16 input-slot row (embedding at slots 10..65)
    h0_tab[(size_t)t * 96 + threadIdx.x] = (__bf16)((threadIdx.x >= 10 && threadIdx.x < 10 + NEMB) ? emb[threadIdx.x - 10] : 0.0f);
  const int o = threadIdx.x;
  if (o < 96) {
    float acc = 0.0f;
    if (o < D) {
      const float* W = params + o * D + 9;
      acc = params[D * D + o];
#pragma unroll 8
      for (int e = 0; e < NEMB; e++) acc = fmaf(W[e], emb[e], acc);
    }
    beff[(size_t)t * 96 + o] = fmaf(scale, acc, offset);  // the table coordinate 16 z + 127.5 when the SiLU table is on (so3x_mlp.hpp)
  }
}

// ---- prep: per-timestep layer-0 fragments of the bf16 chain (layer0_chain_t) ------------------------------
// element j of lane (i, h) of tile `to` feeds K slot 8h + j:  slots 0..8 = 16 W_0[o][slot], 9 / 10 = bf16 halves of
// 16 beff[t][o], 11 = the table offset 127.5 (k_prep wrote beff as 16 beff + 127.5), o = 32 to + i.
__global__ void __launch_bounds__(192) k_prep_l0t(const float* __restrict__ params, const float* __restrict__ beff, void* __restrict__ l0t,
                                                  int t_first, int f16) {
  const int t = t_first + blockIdx.x, to = threadIdx.x >> 6, lane = threadIdx.x & 63, i = lane & 31, h = lane >> 5;
  const int o = 32 * to + i;
  const float be = beff[(size_t)t * 96 + o] - kTabD;
  const float hi = f16 ? (float)(_Float16)be : (float)(__bf16)be;   // the bias as value + remainder in two K slots
  float val8[8];
#pragma unroll
  for (int j = 0; j < 8; j++) {
    const int slot = 8 * h + j;
    float val = 0.0f;
    if (slot < 9) val = o < D ? kTabC * params[o * D + slot] : 0.0f;
    else if (slot == 9) val = hi;
    else if (slot == 10) val = be - hi;
    else if (slot == 11) val = kTabD;
    val8[j] = val;
  }
  reinterpret_cast<bf16x8*>(l0t)[((size_t)t * 3 + to) * 64 + lane] = f16 ? pack_octet<true>(val8) : pack_octet<false>(val8);
}

// ---- standalone forward ---------------------------------------------------------------
// 4 waves per block; each wave walks 32-sample tiles.  Both lanes of a sample column read
// the sample's 9 rotation entries straight from global (36-B stride; compute-bound kernel).
template <int PREC, int VAR>
__global__ void __launch_bounds__(256, 2)
k_mlp_fwd(const void* __restrict__ gimg, const float* __restrict__ beff_tab, const float* __restrict__ R,
          const int64_t* __restrict__ t, int64_t t_stride, Freqs fr, float* __restrict__ out, int64_t n, int nout) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  load_image(gimg, lds, image_bytes<PREC, VAR>());
  __syncthreads();
  const int lane = threadIdx.x & 63, col = lane & 31, h = lane >> 5;
  const int64_t ntiles = (n + 31) / 32;
  const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
  for (int64_t tile = wave; tile < ntiles; tile += nwaves) {
    int64_t idx = tile * 32 + col;
    const bool live = idx < n;
    if (!live) idx = n - 1;
    float x[9];
    load_rot9(R, idx, x);
    const int64_t tt = t[idx * t_stride];
    float v[NOUT_MAX];
    forward_tile<PREC, VAR, 0, false, NOUT_MAX>(lds, x, chain_layout(VAR) ? beff_tab + (size_t)tt * 96 : nullptr, tt, &fr, v, lane);
    if (live && h == 0) {
      out[idx * nout] = v[0]; out[idx * nout + 1] = v[1]; out[idx * nout + 2] = v[2];
      if (nout == 6) { out[idx * 6 + 3] = v[3]; out[idx * 6 + 4] = v[4]; out[idx * 6 + 5] = v[5]; }
    }
  }
}

template <int PREC, int VAR> int launch_prep_t(hipStream_t s, const float* params, int T, void* ws, int nout, void* wt, bool want_image,
                                               unsigned* zero_word, int t_first, int t_count, bool f16 = false) {
  const bool tables = chain_layout(VAR) && T > 0;
  float* beff = tables ? reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + beff_offset(PREC, VAR)) : nullptr;
  float* emb = (tables && gather_layout(VAR)) ? reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + emb_offset(PREC, VAR, T)) : nullptr;
  __bf16* h0 = (tables && gather_layout(VAR)) ? reinterpret_cast<__bf16*>(reinterpret_cast<char*>(ws) + h0_offset(PREC, VAR, T)) : nullptr;
  const int rows = !tables ? 0 : (t_count > 0 ? t_count : T);
  hipLaunchKernelGGL((k_prep<PREC, VAR>), dim3(PREP_IMG_BLOCKS + PREP_WT_BLOCKS + rows), dim3(256), 0, s, params,
                     want_image ? ws : nullptr, wt, nout, host_freqs(), T, fold_scale<PREC, VAR>() ? kTabC : 1.0f,
                     fold_scale<PREC, VAR>() ? kTabD : 0.0f, beff, emb, h0, zero_word, t_count > 0 ? t_first : 0, f16 ? 1 : 0);
  return check_launch();
}

template <int PREC, int VAR>
int launch_fwd_t(hipStream_t s, const void* ws, const float* R, const int64_t* t, int64_t t_stride, float* out, int64_t n, int nout) {
  constexpr int IMG = image_bytes<PREC, VAR>();
  static PerDevice attr;
  if (int rc = ensure_dyn_lds(attr, reinterpret_cast<const void*>(&k_mlp_fwd<PREC, VAR>), IMG)) return rc;
  const int64_t ntiles = (n + 31) / 32;
  const int64_t want = (ntiles + 3) / 4;
  const int max_blocks = IMG > 80 * 1024 ? 256 : 512;  // LDS-limited residency: 1 or 2 blocks per CU
  const int grid = (int)(want < max_blocks ? want : max_blocks);
  const float* beff = chain_layout(VAR) ? reinterpret_cast<const float*>(reinterpret_cast<const char*>(ws) + beff_offset(PREC, VAR))
                                        : nullptr;
  hipLaunchKernelGGL((k_mlp_fwd<PREC, VAR>), dim3(grid), dim3(256), IMG, s, ws, beff, R, t, t_stride, host_freqs(), out, n, nout);
  return check_launch();
}

}  // namespace

namespace so3x {
namespace mlp {
// after launch_prep(bf16, CHAIN, T): the per-timestep layer-0 fragments of the chain kernel, workspace >= l0t_end(T)
int launch_prep_l0t(hipStream_t s, const float* params, int T, void* workspace, int t_first, int t_count, bool f16) {
  char* ws = reinterpret_cast<char*>(workspace);
  hipLaunchKernelGGL(k_prep_l0t, dim3(t_count > 0 ? t_count : T), dim3(192), 0, s, params,
                     reinterpret_cast<const float*>(ws + beff_offset(SO3X_PREC_BF16, CHAIN)), (void*)(ws + l0t_offset(T)),
                     t_count > 0 ? t_first : 0, f16 ? 1 : 0);
  return check_launch();
}
int launch_prep(hipStream_t s, const float* params, int precision, int variant, int T, void* workspace, int nout, void* wt,
                bool want_image, unsigned* zero_word, int t_first, int t_count, bool f16) {
  if (precision == SO3X_PREC_F32) {
    if (variant == CHAIN) return launch_prep_t<SO3X_PREC_F32, CHAIN>(s, params, T, workspace, nout, wt, want_image, zero_word, t_first, t_count);
    if (variant == GATHER) return launch_prep_t<SO3X_PREC_F32, GATHER>(s, params, T, workspace, nout, wt, want_image, zero_word, t_first, t_count);
    return launch_prep_t<SO3X_PREC_F32, FULL>(s, params, T, workspace, nout, wt, want_image, zero_word, t_first, t_count);
  }
  if (variant == CHAIN) return launch_prep_t<SO3X_PREC_BF16, CHAIN>(s, params, T, workspace, nout, wt, want_image, zero_word, t_first, t_count, f16);
  if (variant == GATHER) return launch_prep_t<SO3X_PREC_BF16, GATHER>(s, params, T, workspace, nout, wt, want_image, zero_word, t_first, t_count);
  if (variant == GATHER_T) return launch_prep_t<SO3X_PREC_BF16, GATHER_T>(s, params, T, workspace, nout, wt, want_image, zero_word, t_first, t_count);
  if (variant == GATHER_TD) return launch_prep_t<SO3X_PREC_BF16, GATHER_TD>(s, params, T, workspace, nout, wt, want_image, zero_word, t_first, t_count);
  return launch_prep_t<SO3X_PREC_BF16, FULL>(s, params, T, workspace, nout, wt, want_image, zero_word, t_first, t_count);
}
}  // namespace mlp
}  // namespace so3x

extern "C" {

int so3x_mlp_fwd(so3x_stream_t s, const float* params, const float* R, const int64_t* t, int64_t t_stride, float* out,
                 int64_t n, int n_out, int precision, int t_table, void* workspace, size_t workspace_bytes) {
  if (n < 0 || (n && (!params || !R || !t || !out)) || (t_stride != 0 && t_stride != 1) || t_table < 0 ||
      (n_out != 3 && n_out != 6))
    return SO3X_ERR_INVALID_ARG;
  if (precision != SO3X_PREC_F32 && precision != SO3X_PREC_BF16) return SO3X_ERR_UNSUPPORTED;
  if (!workspace || workspace_bytes < (t_table ? tables_end(precision, CHAIN, t_table) : image_bytes_rt(precision, FULL)))
    return SO3X_ERR_WORKSPACE;
  if (n == 0) return SO3X_OK;
  if (t_table > 0) {  // bounded timesteps: per-timestep effective-bias rows gathered per sample, no in-kernel sin/cos
    int rc = launch_prep((hipStream_t)s, params, precision, CHAIN, t_table, workspace, n_out);
    if (rc) return rc;
    if (precision == SO3X_PREC_F32) return launch_fwd_t<SO3X_PREC_F32, CHAIN>((hipStream_t)s, workspace, R, t, t_stride, out, n, n_out);
    return launch_fwd_t<SO3X_PREC_BF16, CHAIN>((hipStream_t)s, workspace, R, t, t_stride, out, n, n_out);
  }
  int rc = launch_prep((hipStream_t)s, params, precision, FULL, 0, workspace, n_out);
  if (rc) return rc;
  if (precision == SO3X_PREC_F32) return launch_fwd_t<SO3X_PREC_F32, FULL>((hipStream_t)s, workspace, R, t, t_stride, out, n, n_out);
  return launch_fwd_t<SO3X_PREC_BF16, FULL>((hipStream_t)s, workspace, R, t, t_stride, out, n, n_out);
}

}  // extern "C"
