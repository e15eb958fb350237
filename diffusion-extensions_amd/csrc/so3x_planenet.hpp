// so3x_planenet.hpp -- parameter / workspace layout of the PlaneNet denoiser (reference models.py:185-210) shared by the
// exact-fp32 form (so3x_planenet.hip) and the bf16 matrix-core form (so3x_planenet_bf16.hip).
#pragma once
#include "so3x_common.hpp"

namespace so3x {
namespace plane {

// shape of one network
struct Shape {
  int64_t B, P;   // clouds, points per cloud
  int d, H, L, F; // model width, heads, encoder layers, feed-forward width
  int64_t N() const { return B * P; }
  int dh() const { return d / H; }   // head width
  int d2() const { return d / 2; }   // SIREN / time-embedding width
};

// Dropout of a training-mode forward (nn.TransformerEncoderLayer's default p = 0.1, which the reference trains with:
// aircraft_rotate.py:66 `net.train()`, models.py:190): four sites per layer -- the attention probabilities, the attention block's
// output, the feed-forward's hidden activations, the feed-forward's output (torch/nn/modules/transformer.py).  Element e of site
// s keeps its value (scaled by 1 / (1 - p)) iff 16-bit piece (e & 7) of Philox4x32-10(key = seed; counter = (e >> 3,
// offset << 8 | s)) -- words x, y, z, w, low half first -- is >= p * 2^16 (eight decisions per call: a lane's eight consecutive
// columns); the backward regenerates the same masks from (seed, offset).  torch's own mask stream cannot be reproduced
// (it depends on its kernels' launch geometry); parity is against an emulation with THESE masks (tests/test_planenet.py).
struct Drop {
  float p;
  uint64_t seed, offset;
  bool on() const { return p > 0.f; }
  uint32_t thr16() const { const double v = (double)p * 65536.0; return v >= 65535.0 ? 65535u : (uint32_t)v; }
  float inv_keep() const { return 1.f / (1.f - p); }
  uint64_t ctr_hi(int layer, int site) const { return (offset << 8) | (uint64_t)(4 * layer + site); }
};
enum { DROP_ATTN = 0, DROP_BLOCK1 = 1, DROP_FFN = 2, DROP_BLOCK2 = 3 };

inline bool shape_ok(const Shape& s) {
  return s.B >= 0 && s.P >= 1 && s.d >= 8 && s.H >= 1 && s.L >= 1 && s.F >= 1 && s.d % s.H == 0 && s.d % 4 == 0 && s.d <= 4096 &&
         s.F <= 16384 && s.P <= (1 << 20) && s.B * s.P < (int64_t(1) << 31);
}

// Offsets (in floats) into the flat parameter buffer = state_dict order of the reference's module (models.py:186-196):
// encoder.layers.{l}.{self_attn.in_proj_weight, in_proj_bias, self_attn.out_proj.weight, .bias, linear1.weight, .bias,
// linear2.weight, .bias, norm1.weight, .bias, norm2.weight, .bias}, position_siren.{positional.weight, .bias, post_scale.weight,
// .bias}, out_net.0.pool.0.{weight, bias}, out_net.0.lin.{weight, bias}, out_net.1.{weight, bias}.
struct LayerOff { int64_t wqkv, bqkv, wo, bo, w1, b1, w2, b2, g1, be1, g2, be2; };
struct ParamOff {
  int64_t per_layer;
  int64_t wp, bp, wps, bps, wpool, bpool, wlin, blin, wout, bout, total;
  int d, F;
  __host__ __device__ LayerOff layer(int l) const {
    LayerOff o;
    int64_t p = per_layer * l;
    const int64_t dd = (int64_t)d * d;
    o.wqkv = p; p += 3 * dd;
    o.bqkv = p; p += 3 * d;
    o.wo = p; p += dd;
    o.bo = p; p += d;
    o.w1 = p; p += (int64_t)F * d;
    o.b1 = p; p += F;
    o.w2 = p; p += (int64_t)d * F;
    o.b2 = p; p += d;
    o.g1 = p; p += d;
    o.be1 = p; p += d;
    o.g2 = p; p += d;
    o.be2 = p; p += d;
    return o;
  }
};
inline ParamOff param_offsets(const Shape& s) {
  ParamOff o;
  o.d = s.d;
  o.F = s.F;
  const int64_t d = s.d, F = s.F, d2 = s.d / 2;
  o.per_layer = 4 * d * d + 2 * d * F + 9 * d + F;
  int64_t p = o.per_layer * s.L;
  o.wp = p; p += d2 * 3;
  o.bp = p; p += d2;
  o.wps = p; p += d2 * d2;
  o.bps = p; p += d2;
  o.wpool = p; p += d;
  o.bpool = p; p += 1;
  o.wlin = p; p += d * d;
  o.blin = p; p += d;
  o.wout = p; p += 3 * d;
  o.bout = p; p += 3;
  o.total = p;
  return o;
}

constexpr size_t kAlign = 256;
inline size_t up(size_t x) { return (x + kAlign - 1) / kAlign * kAlign; }

// sequential carve of a byte range
struct Carve {
  char* base;
  size_t off = 0;
  explicit Carve(void* b) : base(reinterpret_cast<char*>(b)) {}
  template <typename T>
  T* take(size_t count) {
    T* p = reinterpret_cast<T*>(base + off);
    off += up(count * sizeof(T));
    return p;
  }
};

// ---- the exact-fp32 building blocks (so3x_planenet.hip) the bf16 form uses for its small products and reductions
struct Mat {  // a strided matrix view: element (i, j) at p[i * s0 + j * s1]
  const float* p;
  int64_t s0, s1;
};
inline Mat rowmajor(const float* p, int64_t ld) { return Mat{p, ld, 1}; }
inline Mat transposed(const float* p, int64_t ld) { return Mat{p, 1, ld}; }   // view (i, j) = stored [j][i]
// C[z][m][n] = act(alpha * sum_k A[z](m, k) B[z](k, n) + bias[n] (+ C if accumulate)) on v_mfma_f32_32x32x2_f32; z = (z0 < nb0, z1 < nb1)
int gemm(hipStream_t s, Mat A, Mat B, float* C, int64_t ldc, int M, int N, int K, const float* bias = nullptr, float alpha = 1.f,
         bool relu = false, bool accumulate = false, int nb0 = 1, int nb1 = 1, int64_t sA0 = 0, int64_t sA1 = 0, int64_t sB0 = 0,
         int64_t sB1 = 0, int64_t sC0 = 0, int64_t sC1 = 0);
// the same product (no activation) for a small M x N and a long K, cut over K into slab[chunk][M][N] partials summed in a
// fixed order: what a weight gradient over all tokens needs to fill the chip (falls back to gemm() when K is short)
// (+ bias, + the old C when `accumulate`, both applied by the kernel that sums the chunks)
int gemm_splitk(hipStream_t s, Mat A, Mat B, float* C, int64_t ldc, int M, int N, int K, float* slab, size_t slab_floats,
                const float* bias = nullptr, bool accumulate = false);
// out[c] = sum over rows of X[row][c] (* xhat[row][c] when r / stats are given), fixed order; part: colsum_part_floats(rows, cols) floats
// (rows per chunk: about rows / 256 -- between 32 and 512 -- so that a small matrix still makes ~256 chunks: at the reference's own
//  batch sizes a fixed 512 left a bias gradient to four workgroups whose threads each walked 128 rows one dependent load after the other)
constexpr int CH = 512, COLSUM_NARROW_FLOATS = 65536;
inline int colsum_ch(int64_t rows, int cols = 1 << 30) {
  int64_t c = (rows / 256 + 31) / 32 * 32;
  c = c < 32 ? 32 : (c > CH ? CH : c);
  // a NARROW matrix (a 64-wide bias gradient is one workgroup per chunk) is cut finer: as many chunks as keep chunks x cols within
  // COLSUM_NARROW_FLOATS of partials
  const int64_t narrow = ((rows * (int64_t)cols + COLSUM_NARROW_FLOATS - 1) / COLSUM_NARROW_FLOATS + 31) / 32 * 32;
  const int64_t cn = narrow < 32 ? 32 : narrow;
  return (int)(cn < c ? cn : c);
}
// floats to make room for so that colsum() of ANY matrix of at most `rows` rows and at most `widest` columns fits (a plan sums
// token-level and batch-level matrices through one buffer): by the rows rule at most 264 chunks while rows / 256 <= 512, rows / 512
// beyond, of `widest` floats each; by the narrow rule COLSUM_NARROW_FLOATS plus one chunk
inline size_t colsum_part_floats(int64_t rows, size_t widest) {
  const int64_t large = (rows + CH - 1) / CH;
  const size_t by_rows = (size_t)((large > 288 ? large : 288) + 1) * widest, narrow = (size_t)COLSUM_NARROW_FLOATS + 2 * widest;
  return by_rows > narrow ? by_rows : narrow;
}
int colsum(hipStream_t s, const float* X, int64_t ld, int64_t rows, int cols, float* out, float* part, const float* r = nullptr,
           int64_t ldr = 0, const float* stats = nullptr);
// row kernels (one wave per row): r = a + b, y = LayerNorm(r) gamma + beta, stats = (mean, rstd); its backward; dS = P o (dP - rowsum(dP o P))
// scale in place over dP; df = f > 0 ? df scale : 0; ds *= cos(pre)
// dst[e] = keep(e) ? src[e] / (1 - p) : 0 over a flat array (src may be dst): dropout site `site` of layer `layer` (Drop)
int dropout_apply(hipStream_t s, const Drop& dr, int layer, int site, const float* src, float* dst, int64_t n);
int add_ln(hipStream_t s, const float* a, const float* b, float* r_out, float* y, float* stats, const float* gamma, const float* beta, int64_t N,
           int d, float eps);
int ln_bwd(hipStream_t s, const float* dy, const float* r, const float* stats, const float* gamma, float* dr, int64_t N, int d);
int softmax_bwd(hipStream_t s, const float* probs, float* dprobs, int64_t rows, int cols, float scale);
int relu_bwd(hipStream_t s, float* df, const float* f, int64_t n, float scale);
int cos_mul(hipStream_t s, float* ds, const float* pre, int64_t n);
// pooled[b] = Wlin xs[b] + blin, out[b] = Wout pooled[b] + bout: one workgroup per cloud
int head(hipStream_t s, const float* xs, const float* wlin, const float* blin, const float* wout, const float* bout, float* pooled, float* out,
         int64_t B, int d);

}  // namespace plane
}  // namespace so3x
