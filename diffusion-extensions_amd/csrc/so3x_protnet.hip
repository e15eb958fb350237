// so3x_protnet.hip -- the ProtNet docking denoiser (reference models.py:212-319, driven by prot_train.py:90-108 through
// ProjectedSE3Diffusion, diffusion.py:525-573), forward and backward, as hand-written kernels: EXACT-FP32 form for any
// (dim, heads, t_depth, c_depth) -- the parity form, and the one prot_train.py's own argparse defaults (dim 1024, 8 heads, 12 + 8
// layers) run on.  The bf16 matrix-core form of the class-default width (dim 64, 4 heads) lives in so3x_protnet_bf16.hip.
//
//   per chain (receptor and ligand of every complex: 2 B ragged sequences, handed over CSR-style and padded HERE to Lp rows):
//     res_emb = res_conv(one-hot residues): Conv1d(21 -> d, k 3, pad 1) SiLU [x + SiLU(Conv1d(d -> d))] x (c_depth - 2) Conv1d(d -> d - d/2 - d/4)
//               over the chain's OWN length (zero padding at ITS ends), zero at padded rows             (models.py:226-252, 281-282)
//     pos_emb = Siren(3 -> d/2, scale 0.1)(positions), ang_emb = Siren(9 -> d/4)(frames), padded rows computed from zeros as the
//               reference does                                                                              (models.py:276-280)
//     enc     = rec_tf([res_emb | pos_emb | ang_emb], src_key_padding_mask): t_depth post-norm encoder layers (ReLU, ffn 2048) + the
//               encoder's final LayerNorm; padded keys take no probability mass; BOTH chains through rec_tf (models.py:288, 302:
//               the reference never calls lig_tf -- reproduced, lig_tf's gradient is zero)
//     pool    = PoolRN(enc, mask) [d], PoolPos(enc, positions, mask) [3] with the chain kind's own pool parameters (models.py:94-127)
//   out = last([SinusoidalPosEmb(d)(t) | rec pool | rec pos | lig pool | lig pos]) : Linear(3 d + 6 -> d) SiLU, 3 x [x + SiLU(Linear)],
//         Linear(d -> 6) = (rot_g, shift_g)                                                                 (models.py:261-270, 311-318)
//
// Every product runs on so3x::plane::gemm (v_mfma_f32_32x32x2_f32: bit-for-bit a k-ordered fp32 fmaf chain): the convolutions as
// three row-shifted products over buffers with one zero halo row either side of a chain, attention as (chain, head)-batched
// products whose [S][H][Lp][Lp] probabilities are kept for the backward.  Reductions in fixed orders, no atomics.
#include "so3x_protnet.hpp"
#include "so3x_math.hpp"

namespace so3x {
namespace prot {

using plane::colsum;
using plane::Drop;
using plane::dropout_apply;
using plane::DROP_ATTN;
using plane::DROP_BLOCK1;
using plane::DROP_BLOCK2;
using plane::DROP_FFN;
using plane::gemm;
using plane::gemm_splitk;
using plane::Mat;
using plane::rowmajor;
using plane::transposed;

#define TRY(expr)                 \
  do {                            \
    int rc__ = (expr);            \
    if (rc__) return rc__;        \
  } while (0)

inline unsigned nblk(int64_t n, int per) { return (unsigned)((n + per - 1) / per); }

__device__ __forceinline__ float wsum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float wmax(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ float sigm(float x) { return 1.f / (1.f + expf(-x)); }

// ------------------------------------------------------------------------------------------------ ragged -> padded
// rows [off[c], off[c + 1]) of the receptor / ligand arrays -> chain s = c (+ B for ligands), rows 0 .. len - 1 of Lp; the rest zero
// (pad_sequence, models.py:277-298).  The residue one-hots go into a halo buffer (row l + 1 of Lp + 2).  len[s] is written here.
__global__ __launch_bounds__(256) void k_pack(const float* __restrict__ rres, const float* __restrict__ rpos, const float* __restrict__ rang,
                                              const int64_t* __restrict__ roff, const float* __restrict__ lres, const float* __restrict__ lpos,
                                              const float* __restrict__ lang, const int64_t* __restrict__ loff, int* __restrict__ len,
                                              float* __restrict__ resh, float* __restrict__ pos, float* __restrict__ ang, int64_t B, int64_t Lp) {
  const int64_t n = (int64_t)blockIdx.x * 8 + (threadIdx.x >> 5);   // 32 lanes per token
  if (n >= 2 * B * Lp) return;
  const int j = threadIdx.x & 31;
  const int64_t s = n / Lp, l = n - s * Lp;
  const bool lig = s >= B;
  const int64_t* off = lig ? loff : roff;
  const int64_t c = lig ? s - B : s, o0 = off[c];
  const int64_t L = off[c + 1] - o0;
  if (l == 0 && j == 0) len[s] = (int)(L < Lp ? L : Lp);
  const bool ok = l < L;
  const int64_t src = o0 + l;
  const float* res = lig ? lres : rres;
  const float* ps = lig ? lpos : rpos;
  const float* an = lig ? lang : rang;
  float* rh = resh + (s * (Lp + 2) + l + 1) * RES;
  if (j < RES) rh[j] = ok ? res[src * RES + j] : 0.f;
  if (j < 3) pos[n * 3 + j] = ok ? ps[src * 3 + j] : 0.f;
  if (j >= 8 && j < 17) ang[n * 9 + j - 8] = ok ? an[src * 9 + j - 8] : 0.f;
}

__global__ __launch_bounds__(256) void k_sin(const float* __restrict__ pre, float* __restrict__ sn, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) sn[i] = sinf(pre[i]);
}

// dst[s][l + hd][c] = l < len[s] ? (res ? res[same] : 0) + silu(z[s][l][c]) : 0;  dst / res: chains of Lp + 2 hd rows (hd = 1: halo
// buffers), z: chains of Lp rows.  len == nullptr: no mask (the head's rows).
__global__ __launch_bounds__(256) void k_silu_mask(const float* __restrict__ z, const float* res, float* dst, const int* __restrict__ len,
                                                   int64_t S, int64_t Lp, int C, int hd) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= S * Lp * C) return;
  const int64_t row = i / C, s = row / Lp, l = row - s * Lp;
  const int c = (int)(i - row * C);
  const int64_t o = (s * (Lp + 2 * hd) + l + hd) * C + c;
  float v = 0.f;
  if (!len || l < len[s]) {
    const float zz = z[i];
    v = zz * sigm(zz) + (res ? res[o] : 0.f);
  }
  dst[o] = v;
}
// x[s][l][0 .. C) = 0 for l >= len[s] (x rows of width ld): pad_sequence's zeros at the padded rows of the conv output
__global__ __launch_bounds__(256) void k_mask_rows(float* __restrict__ x, const int* __restrict__ len, int64_t S, int64_t Lp, int C, int ld) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= S * Lp * C) return;
  const int64_t row = i / C, s = row / Lp, l = row - s * Lp;
  if (l >= len[s]) x[row * ld + (i - row * C)] = 0.f;
}
// dzh[s][l + 1][c] (halo) = l < len[s] ? dy[s][l][c (ld)] * (z ? silu'(z[s][l][c]) : 1) : 0
__global__ __launch_bounds__(256) void k_silu_bwd_halo(const float* __restrict__ dy, int ld, const float* z, float* __restrict__ dzh,
                                                       const int* __restrict__ len, int64_t S, int64_t Lp, int C) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= S * Lp * C) return;
  const int64_t row = i / C, s = row / Lp, l = row - s * Lp;
  const int c = (int)(i - row * C);
  float v = 0.f;
  if (l < len[s]) {
    v = dy[row * ld + c];
    if (z) {
      const float zz = z[i], sg = sigm(zz);
      v *= sg * (1.f + zz * (1.f - sg));
    }
  }
  dzh[(s * (Lp + 2) + l + 1) * C + c] = v;
}
// dz = da * silu'(z), flat (the head)
__global__ __launch_bounds__(256) void k_silu_bwd(const float* __restrict__ da, const float* __restrict__ z, float* __restrict__ dz, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float zz = z[i], sg = sigm(zz);
  dz[i] = da[i] * sg * (1.f + zz * (1.f - sg));
}
// dW[co][ci][k] = tmp[k][co][ci]
__global__ __launch_bounds__(256) void k_conv_w_scatter(const float* __restrict__ tmp, float* __restrict__ dW, int cout, int cin) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= cout * cin * 3) return;
  const int k = i % 3, ci = (i / 3) % cin, co = i / (3 * cin);
  dW[i] = tmp[((int64_t)k * cout + co) * cin + ci];
}

// rows of S -> softmax over the chain's unmasked keys (columns < len[s]); masked keys get probability 0 (src_key_padding_mask:
// -inf before the softmax, models.py:288).  One wave per row; rows_per_chain = H Lp.
__global__ __launch_bounds__(256) void k_softmax_masked(float* __restrict__ Sm, const int* __restrict__ len, int64_t rows, int cols, int64_t rows_per_chain) {
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  const int n = len[row / rows_per_chain];
  float* r = Sm + row * cols;
  if (cols <= 512) {   // the row in registers: one read and one write of it (the same operations in the same order as below)
    float v[8];
    float m = -INFINITY;
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const int c = lane + 64 * j;
      v[j] = c < n ? r[c] : 0.f;
      if (c < n) m = fmaxf(m, v[j]);
    }
    m = wmax(m);
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < 8; j++)
      if (lane + 64 * j < n) {
        v[j] = expf(v[j] - m);
        sum += v[j];
      }
    sum = wsum(sum);
    const float inv = 1.f / sum;
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const int c = lane + 64 * j;
      if (c < cols) r[c] = c < n ? v[j] * inv : 0.f;
    }
    return;
  }
  float m = -INFINITY;
  for (int c = lane; c < n; c += 64) m = fmaxf(m, r[c]);
  m = wmax(m);
  float sum = 0.f;
  for (int c = lane; c < n; c += 64) {
    const float e = expf(r[c] - m);
    r[c] = e;
    sum += e;
  }
  sum = wsum(sum);
  const float inv = 1.f / sum;
  for (int c = lane; c < cols; c += 64) r[c] = c < n ? r[c] * inv : 0.f;
}

// SinusoidalPosEmb(d)(t) (models.py:13-25) -> pv[b][0 .. d)
__global__ __launch_bounds__(256) void k_time_emb(const int64_t* __restrict__ t, float* __restrict__ pv, int64_t B, int d, int pw, float neg_emb) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= B * d) return;
  const int64_t b = i / d;
  const int j = (int)(i - b * d), half = d / 2, jj = j < half ? j : j - half;
  const float f = (float)exp((double)((float)jj * neg_emb));
  const float arg = (float)t[b] * f;
  pv[b * pw + j] = j < half ? sinf(arg) : cosf(arg);
}

// PoolRN's weighted mean and PoolPos (models.py:94-127) of one chain per workgroup, with the chain kind's own parameters:
//   wE_l = sigmoid(x_l . wpool + bpool) [l < len],  xs = sum wE_l x_l / max(sum wE_l, 1e-6)      (lin is applied to xs afterwards:
//   wP_l = sigmoid(x_l . wppool + bppool) [l < len], pos_out = sum wP_l pos_l / max(sum wP_l, 1e-6)   it is linear)
// rows in the order l = wave, wave + 4, ...; the four waves' partial sums are added in a fixed order.
constexpr int PMAX = 16;   // d <= 1024
__global__ __launch_bounds__(256) void k_pool(const float* __restrict__ enc, const float* __restrict__ pos, const int* __restrict__ len,
                                              const float* __restrict__ prm, Pool rec, Pool lig, float* __restrict__ wE, float* __restrict__ wP,
                                              float* __restrict__ SE, float* __restrict__ SP, float* __restrict__ xs, float* __restrict__ pv,
                                              int64_t B, int64_t Lp, int d, int pw) {
  extern __shared__ float sm[];   // [4][d] + [4][8]
  const int64_t s = blockIdx.x;
  const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
  const Pool q = s >= B ? lig : rec;
  const float* we = prm + q.wpool;
  const float* wp = prm + q.wppool;
  const float be = prm[q.bpool], bp = prm[q.bppool];
  const int n = len[s];
  float acc[PMAX];
#pragma unroll
  for (int i = 0; i < PMAX; i++) acc[i] = 0.f;
  float se = 0.f, sp = 0.f, p0 = 0.f, p1 = 0.f, p2 = 0.f;
  for (int l = g; l < Lp; l += 4) {
    const int64_t row = s * Lp + l;
    if (l >= n) {
      if (lane == 0) wE[row] = wP[row] = 0.f;
      continue;
    }
    const float* x = enc + row * d;
    float a = 0.f, b = 0.f;
    for (int c = lane; c < d; c += 64) {
      a = fmaf(x[c], we[c], a);
      b = fmaf(x[c], wp[c], b);
    }
    const float e = sigm(wsum(a) + be), f = sigm(wsum(b) + bp);
#pragma unroll
    for (int i = 0; i < PMAX; i++)
      if (lane + 64 * i < d) acc[i] = fmaf(e, x[lane + 64 * i], acc[i]);
    se += e;
    sp += f;
    p0 = fmaf(f, pos[row * 3 + 0], p0);
    p1 = fmaf(f, pos[row * 3 + 1], p1);
    p2 = fmaf(f, pos[row * 3 + 2], p2);
    if (lane == 0) {
      wE[row] = e;
      wP[row] = f;
    }
  }
#pragma unroll
  for (int i = 0; i < PMAX; i++)
    if (lane + 64 * i < d) sm[g * d + lane + 64 * i] = acc[i];
  float* tail = sm + 4 * d + g * 8;
  if (lane == 0) {
    tail[0] = se;
    tail[1] = sp;
    tail[2] = p0;
    tail[3] = p1;
    tail[4] = p2;
  }
  __syncthreads();
  const float* t0 = sm + 4 * d;
  const float tse = (t0[0] + t0[8]) + (t0[16] + t0[24]), tsp = (t0[1] + t0[9]) + (t0[17] + t0[25]);
  const float ce = fmaxf(tse, 1e-6f), cp = fmaxf(tsp, 1e-6f);
  for (int c = threadIdx.x; c < d; c += 256) xs[s * d + c] = ((sm[c] + sm[d + c]) + (sm[2 * d + c] + sm[3 * d + c])) / ce;
  if (threadIdx.x < 3) {
    const int j = threadIdx.x + 2;
    const int64_t b = s >= B ? s - B : s;
    pv[b * pw + (s >= B ? 3 * d + 3 : 2 * d) + threadIdx.x] = ((t0[j] + t0[8 + j]) + (t0[16 + j] + t0[24 + j])) / cp;
  }
  if (threadIdx.x == 0) {
    SE[s] = tse;
    SP[s] = tsp;
  }
}
// token-level backward of both pools: with dxs = d / d xs (through lin) and dpo = d / d pos_out,
//   e_l = (x_l . dxs - [SE >= 1e-6] xs . dxs) / max(SE, 1e-6),  gE_l = e_l wE_l (1 - wE_l)
//   f_l = (pos_l . dpo - [SP >= 1e-6] pos_out . dpo) / max(SP, 1e-6),  gP_l = f_l wP_l (1 - wP_l)
//   dx_l = (wE_l / max(SE, 1e-6)) dxs + gE_l wpool + gP_l wppool       (zero at masked rows: wE = wP = 0 there)
__global__ __launch_bounds__(256) void k_pool_bwd(const float* __restrict__ enc, const float* __restrict__ pos, const float* __restrict__ prm, Pool rec,
                                                  Pool lig, const float* __restrict__ wE, const float* __restrict__ wP, const float* __restrict__ SE,
                                                  const float* __restrict__ SP, const float* __restrict__ xs, const float* __restrict__ pv,
                                                  const float* __restrict__ dxs, const float* __restrict__ dpv, float* __restrict__ dx,
                                                  float* __restrict__ gE, float* __restrict__ gP, int64_t B, int64_t Lp, int d, int pw) {
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= 2 * B * Lp) return;
  const int lane = threadIdx.x & 63;
  const int64_t s = row / Lp, b = s >= B ? s - B : s;
  const Pool q = s >= B ? lig : rec;
  const int pcol = s >= B ? 3 * d + 3 : 2 * d;
  const float e = wE[row], f = wP[row];
  float* o = dx + row * d;
  if (e == 0.f && f == 0.f) {   // a masked row
    for (int c = lane; c < d; c += 64) o[c] = 0.f;
    if (lane == 0) gE[row] = gP[row] = 0.f;
    return;
  }
  const float* x = enc + row * d;
  const float* dxb = dxs + s * d;
  float a = 0.f, c0 = 0.f;
  for (int c = lane; c < d; c += 64) {
    a = fmaf(x[c], dxb[c], a);
    c0 = fmaf(xs[s * d + c], dxb[c], c0);
  }
  a = wsum(a);
  c0 = wsum(c0);
  const float se = SE[s], ce = fmaxf(se, 1e-6f), sp = SP[s], cp = fmaxf(sp, 1e-6f);
  const float ge = (a - (se >= 1e-6f ? c0 : 0.f)) / ce * e * (1.f - e);
  const float* dpo = dpv + b * pw + pcol;
  const float* po = pv + b * pw + pcol;
  const float pd0 = pos[row * 3] * dpo[0] + pos[row * 3 + 1] * dpo[1] + pos[row * 3 + 2] * dpo[2];
  const float pc0 = po[0] * dpo[0] + po[1] * dpo[1] + po[2] * dpo[2];
  const float gp = (pd0 - (sp >= 1e-6f ? pc0 : 0.f)) / cp * f * (1.f - f);
  const float* we = prm + q.wpool;
  const float* wp = prm + q.wppool;
  for (int c = lane; c < d; c += 64) o[c] = (e / ce) * dxb[c] + ge * we[c] + gp * wp[c];
  if (lane == 0) {
    gE[row] = ge;
    gP[row] = gp;
  }
}

// ------------------------------------------------------------------------------------------------ buffers
struct LayerActs { float *qkv, *probs, *o, *r1, *st1, *x1, *f, *r2, *st2; };
struct Acts {
  int* len;
  float *resh, *pos, *ang, *prep, *snp, *prea, *sna;
  float* xc[34];   // xc[i]: input of conv i (halo layout), i = 1 .. Cd - 1 (xc[0] = resh)
  float* zc[34];   // zc[i]: output of conv i before its SiLU, i = 0 .. Cd - 2
  float* h[66];
  LayerActs layer[65];
  float *enc, *stF, *wE, *wP, *SE, *SP, *xs, *pv;
  float* hz[4];    // the head's pre-activations
  float* ha[4];    // and activations a_0 .. a_3
  float* slab;     // split-K partials of the feed-forward block's second product (plane::gemm_splitk) when the batch is small
  size_t slab_floats;
  size_t bytes;
};
// per_layer: every encoder layer keeps its own buffers (what the backward reads); else the layers share one set and h ping-pongs
inline Acts carve_acts(const Dims& s, void* mem, bool per_layer) {
  Acts a;
  Carve c(mem);
  const size_t N = (size_t)s.N(), R = (size_t)s.R(), S = (size_t)s.S(), d = s.d, B = (size_t)s.B;
  a.len = c.take<int>(S);
  a.resh = c.take<float>(R * RES);
  a.pos = c.take<float>(N * 3);
  a.ang = c.take<float>(N * 9);
  a.prep = c.take<float>(N * s.pd());
  a.snp = c.take<float>(N * s.pd());
  a.prea = c.take<float>(N * s.ad());
  a.sna = c.take<float>(N * s.ad());
  a.xc[0] = a.resh;
  for (int i = 1; i < s.Cd; i++) a.xc[i] = c.take<float>(R * d);
  for (int i = 0; i < s.Cd - 1; i++) a.zc[i] = c.take<float>(N * d);
  if (per_layer) {
    for (int l = 0; l <= s.T; l++) a.h[l] = c.take<float>(N * d);
  } else {
    float* h0 = c.take<float>(N * d);
    float* h1 = c.take<float>(N * d);
    for (int l = 0; l <= s.T; l++) a.h[l] = (l & 1) ? h1 : h0;
  }
  for (int l = 0; l < s.T; l++) {
    if (l == 0 || per_layer) {
      LayerActs& k = a.layer[l];
      k.qkv = c.take<float>(N * 3 * d);
      k.probs = c.take<float>(S * s.H * s.Lp * s.Lp);
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
  a.enc = c.take<float>(N * d);
  a.stF = c.take<float>(N * 2);
  a.wE = c.take<float>(N);
  a.wP = c.take<float>(N);
  a.SE = c.take<float>(S);
  a.SP = c.take<float>(S);
  a.xs = c.take<float>(S * d);
  a.pv = c.take<float>(B * s.pw());
  for (int i = 0; i < 4; i++) {
    a.hz[i] = c.take<float>(B * d);
    a.ha[i] = c.take<float>(B * d);
  }
  // gemm_splitk cuts K only while tiles x chunks <= 1024, i.e. chunks x M x N <= 1024 x 64 x 64 floats -- and not at all from 512
  // row tiles on: nothing to reserve for a large batch
  a.slab_floats = s.N() * (size_t)d < ((size_t)4 << 20) ? ((size_t)4 << 20) : 0;
  a.slab = c.take<float>(a.slab_floats);
  a.bytes = c.off;
  return a;
}

struct BwdBufs {
  float *dA, *dB, *dF, *dqkv, *dO, *dprobs, *dsp, *dsa, *dzh, *dxc, *gE, *gP, *dxs, *dpv, *dh0, *dh1, *dhz, *wtmp, *part, *slab;
  size_t bytes, slab_floats;
};
inline BwdBufs carve_bwd(const Dims& s, void* mem) {
  BwdBufs b;
  Carve c(mem);
  const size_t N = (size_t)s.N(), R = (size_t)s.R(), S = (size_t)s.S(), d = s.d, B = (size_t)s.B;
  b.dA = c.take<float>(N * d);
  b.dB = c.take<float>(N * d);
  b.dF = c.take<float>(N * s.F);
  b.dqkv = c.take<float>(N * 3 * d);
  b.dO = c.take<float>(N * d);
  b.dprobs = c.take<float>(S * s.H * s.Lp * s.Lp);
  b.dsp = c.take<float>(N * s.pd());
  b.dsa = c.take<float>(N * s.ad());
  b.dzh = c.take<float>(R * d);
  b.dxc = c.take<float>(N * d);
  b.gE = c.take<float>(N);
  b.gP = c.take<float>(N);
  b.dxs = c.take<float>(S * d);
  b.dpv = c.take<float>(B * s.pw());
  b.dh0 = c.take<float>(B * d);
  b.dh1 = c.take<float>(B * d);
  b.dhz = c.take<float>(B * d);
  b.wtmp = c.take<float>((size_t)3 * d * (d > RES ? d : RES));
  const size_t widest = (size_t)(s.F > 3 * s.d ? s.F : 3 * s.d);
  b.part = c.take<float>(plane::colsum_part_floats(s.R(), widest));
  // split-K partials of the weight gradients (plane::gemm_splitk): room for 32 chunks of the largest one, at least 8 M floats
  b.slab_floats = (size_t)32 * s.F * d > ((size_t)8 << 20) ? (size_t)32 * s.F * d : ((size_t)8 << 20);
  b.slab = c.take<float>(b.slab_floats);
  b.bytes = c.off;
  return b;
}

// ------------------------------------------------------------------------------------------------ forward plan
// Conv1d(k 3, pad 1) of every chain: z[s][l] = b + sum_k W[:, :, k] x[s][l + k - 1] with x in a halo buffer (row l + k of Lp + 2).
static int conv_fwd(hipStream_t st, const Dims& s, const float* xh, int cin, const float* W, const float* b, float* z, int ldz, int cout) {
  for (int k = 0; k < 3; k++)
    TRY(gemm(st, Mat{xh + (int64_t)k * cin, cin, 1}, Mat{W + k, 3, (int64_t)cin * 3}, z, ldz, (int)s.Lp, cout, cin, k == 0 ? b : nullptr, 1.f, false,
             k > 0, (int)s.S(), 1, (s.Lp + 2) * cin, 0, 0, 0, s.Lp * ldz, 0));
  return SO3X_OK;
}

int forward_f32(hipStream_t st, const Dims& s, const float* prm, const float* rres, const float* rpos, const float* rang, const int64_t* roff,
                const float* lres, const float* lpos, const float* lang, const int64_t* loff, const int64_t* t, float* out, float* pool_out,
                float* enc_out, const Acts& a, const Drop& dr, float* pdrop) {
  const POff po = param_offsets(s);
  const int64_t N = s.N(), S = s.S(), Lp = s.Lp, B = s.B;
  const int d = s.d, H = s.H, dh = s.dh(), F = s.F, pd = s.pd(), ad = s.ad(), rd = s.rd(), pw = s.pw();
  // halo rows must be zero: resh and every conv input buffer
  hipError_t e = hipMemsetAsync(a.resh, 0, (size_t)s.R() * RES * sizeof(float), st);
  if (e != hipSuccess) return (int)e;
  for (int i = 1; i < s.Cd; i++)
    if ((e = hipMemsetAsync(a.xc[i], 0, (size_t)s.R() * d * sizeof(float), st)) != hipSuccess) return (int)e;
  hipLaunchKernelGGL(k_pack, dim3(nblk(N, 8)), dim3(256), 0, st, rres, rpos, rang, roff, lres, lpos, lang, loff, a.len, a.resh, a.pos, a.ang, B, Lp);
  TRY(check_launch());
  float* h0 = a.h[0];
  // residue convolutions -> h0[:, 0 .. rd)
  for (int i = 0; i < s.Cd; i++) {
    const bool last = i == s.Cd - 1;
    TRY(conv_fwd(st, s, a.xc[i], po.cin[i], prm + po.cw[i], prm + po.cb[i], last ? h0 : a.zc[i], last ? d : d, po.cout[i]));
    if (last) {
      hipLaunchKernelGGL(k_mask_rows, dim3(nblk(N * rd, 256)), dim3(256), 0, st, h0, a.len, S, Lp, rd, d);
    } else {
      hipLaunchKernelGGL(k_silu_mask, dim3(nblk(N * d, 256)), dim3(256), 0, st, a.zc[i], i == 0 ? nullptr : a.xc[i], a.xc[i + 1], a.len, S, Lp, d, 1);
    }
    TRY(check_launch());
  }
  // SIREN embeddings (padded rows from zero inputs, as the reference) -> h0[:, rd .. rd + pd), h0[:, rd + pd .. d)
  TRY(gemm(st, rowmajor(a.pos, 3), transposed(prm + po.wpp, 3), a.prep, pd, (int)N, pd, 3, prm + po.bpp));
  hipLaunchKernelGGL(k_sin, dim3(nblk(N * pd, 256)), dim3(256), 0, st, a.prep, a.snp, N * pd);
  TRY(gemm(st, rowmajor(a.snp, pd), transposed(prm + po.wpps, pd), h0 + rd, d, (int)N, pd, pd, prm + po.bpps));
  TRY(gemm(st, rowmajor(a.ang, 9), transposed(prm + po.wap, 9), a.prea, ad, (int)N, ad, 9, prm + po.bap));
  hipLaunchKernelGGL(k_sin, dim3(nblk(N * ad, 256)), dim3(256), 0, st, a.prea, a.sna, N * ad);
  TRY(gemm(st, rowmajor(a.sna, ad), transposed(prm + po.waps, ad), h0 + rd + pd, d, (int)N, ad, ad, prm + po.baps));
  // rec_tf for both chain kinds
  const float scale = 1.f / sqrtf((float)dh);
  for (int l = 0; l < s.T; l++) {
    const LayerOff lo = po.layer(po.rec_tf, l);
    const LayerActs& k = a.layer[l];
    const float* h = a.h[l];
    TRY(gemm(st, rowmajor(h, d), transposed(prm + lo.wqkv, d), k.qkv, 3 * d, (int)N, 3 * d, d, prm + lo.bqkv));
    TRY(gemm(st, rowmajor(k.qkv, 3 * d), transposed(k.qkv + d, 3 * d), k.probs, Lp, (int)Lp, (int)Lp, dh, nullptr, scale, false, false, (int)S, H,
             Lp * 3 * d, dh, Lp * 3 * d, dh, (int64_t)H * Lp * Lp, Lp * Lp));
    hipLaunchKernelGGL(k_softmax_masked, dim3(nblk(S * H * Lp, 4)), dim3(256), 0, st, k.probs, a.len, S * H * Lp, (int)Lp, (int64_t)H * Lp);
    TRY(check_launch());
    // dr.on(): the training-mode forward -- nn.TransformerEncoderLayer's four dropout sites (so3x.h); `pdrop` holds the layer's
    // dropped-out probabilities (the stash keeps the plain softmax, which its backward needs at dropped positions too)
    const float* pv = k.probs;
    if (dr.on()) {
      TRY(dropout_apply(st, dr, l, DROP_ATTN, k.probs, pdrop, S * H * Lp * Lp));
      pv = pdrop;
    }
    TRY(gemm(st, rowmajor(pv, Lp), rowmajor(k.qkv + 2 * d, 3 * d), k.o, d, (int)Lp, dh, (int)Lp, nullptr, 1.f, false, false, (int)S, H,
             (int64_t)H * Lp * Lp, Lp * Lp, Lp * 3 * d, dh, Lp * d, dh));
    TRY(gemm(st, rowmajor(k.o, d), transposed(prm + lo.wo, d), k.r1, d, (int)N, d, d, prm + lo.bo));
    if (dr.on()) TRY(dropout_apply(st, dr, l, DROP_BLOCK1, k.r1, k.r1, N * d));
    TRY(plane::add_ln(st, h, k.r1, k.r1, k.x1, k.st1, prm + lo.g1, prm + lo.be1, N, d, 1e-5f));
    TRY(gemm(st, rowmajor(k.x1, d), transposed(prm + lo.w1, d), k.f, F, (int)N, F, d, prm + lo.b1, 1.f, true));
    if (dr.on()) TRY(dropout_apply(st, dr, l, DROP_FFN, k.f, k.f, N * F));   // (the stash holds the dropped-out activations: what linear2 saw)
    TRY(plane::gemm_splitk(st, rowmajor(k.f, F), transposed(prm + lo.w2, F), k.r2, d, (int)N, d, F, a.slab, a.slab_floats, prm + lo.b2));
    if (dr.on()) TRY(dropout_apply(st, dr, l, DROP_BLOCK2, k.r2, k.r2, N * d));
    TRY(plane::add_ln(st, k.x1, k.r2, k.r2, a.h[l + 1], k.st2, prm + lo.g2, prm + lo.be2, N, d, 1e-5f));
  }
  TRY(plane::add_ln(st, a.h[s.T], nullptr, nullptr, a.enc, a.stF, prm + po.rec_tf.gF, prm + po.rec_tf.bF, N, d, 1e-5f));   // encoder.norm
  if (enc_out && (e = hipMemcpyAsync(enc_out, a.enc, (size_t)N * d * sizeof(float), hipMemcpyDeviceToDevice, st)) != hipSuccess) return (int)e;
  // pools + time embedding -> the [B][3 d + 6] head input
  const float neg_emb = (float)(-(log(10000.0) / (d / 2 - 1)));
  hipLaunchKernelGGL(k_time_emb, dim3(nblk(B * d, 256)), dim3(256), 0, st, t, a.pv, B, d, pw, neg_emb);
  hipLaunchKernelGGL(k_pool, dim3((unsigned)S), dim3(256), (4 * d + 32) * sizeof(float), st, a.enc, a.pos, a.len, prm, po.rec, po.lig, a.wE, a.wP, a.SE,
                     a.SP, a.xs, a.pv, B, Lp, d, pw);
  TRY(check_launch());
  TRY(gemm(st, rowmajor(a.xs, d), transposed(prm + po.rec.wlin, d), a.pv + d, pw, (int)B, d, d, prm + po.rec.blin));
  TRY(gemm(st, rowmajor(a.xs + B * d, d), transposed(prm + po.lig.wlin, d), a.pv + 2 * d + 3, pw, (int)B, d, d, prm + po.lig.blin));
  if (pool_out && (e = hipMemcpyAsync(pool_out, a.pv, (size_t)B * pw * sizeof(float), hipMemcpyDeviceToDevice, st)) != hipSuccess) return (int)e;
  // last: Linear SiLU, 3 x [x + SiLU(Linear(x))], Linear
  TRY(gemm(st, rowmajor(a.pv, pw), transposed(prm + po.w0, pw), a.hz[0], d, (int)B, d, pw, prm + po.b0));
  hipLaunchKernelGGL(k_silu_mask, dim3(nblk(B * d, 256)), dim3(256), 0, st, a.hz[0], nullptr, a.ha[0], nullptr, 1, B, d, 0);
  for (int i = 0; i < 3; i++) {
    TRY(gemm(st, rowmajor(a.ha[i], d), transposed(prm + po.wr[i], d), a.hz[i + 1], d, (int)B, d, d, prm + po.br[i]));
    // (dst = res + silu(z) with res read at dst's own index: stage a_i into a_{i+1} first)
    if ((e = hipMemcpyAsync(a.ha[i + 1], a.ha[i], (size_t)B * d * sizeof(float), hipMemcpyDeviceToDevice, st)) != hipSuccess) return (int)e;
    hipLaunchKernelGGL(k_silu_mask, dim3(nblk(B * d, 256)), dim3(256), 0, st, a.hz[i + 1], a.ha[i + 1], a.ha[i + 1], nullptr, 1, B, d, 0);
  }
  TRY(check_launch());
  TRY(gemm(st, rowmajor(a.ha[3], d), transposed(prm + po.wout, d), out, 6, (int)B, 6, d, prm + po.bout));
  return SO3X_OK;
}

// ------------------------------------------------------------------------------------------------ backward plan
// gradient of conv i: dzh = d z (halo layout, zero at padded rows); dW / db written, dx (optional, [N][cin]) accumulated or written
static int conv_bwd(hipStream_t st, const Dims& s, const float* xh, int cin, const float* W, int cout, const float* dzh, float* dW, float* db,
                    float* dx, bool accumulate_dx, const BwdBufs& w) {
  const int64_t R = s.R();
  for (int k = 0; k < 3; k++)   // dW[:, :, k] = sum over rows r of dz[r] (x) x[r + k - 1] (rows 0 and R - 1 of dzh are halo rows: zero)
    TRY(gemm_splitk(st, transposed(dzh + cout, cout), Mat{xh + (int64_t)k * cin, cin, 1}, w.wtmp + (int64_t)k * cout * cin, cin, cout, cin, (int)(R - 2),
                    w.slab, w.slab_floats));
  hipLaunchKernelGGL(k_conv_w_scatter, dim3(nblk((int64_t)cout * cin * 3, 256)), dim3(256), 0, st, w.wtmp, dW, cout, cin);
  TRY(check_launch());
  TRY(colsum(st, dzh + cout, cout, R - 2, cout, db, w.part));
  if (dx)
    for (int k = 0; k < 3; k++)   // dx[l] = sum_k W[:, :, k]^T dz[l - k + 1]
      TRY(gemm(st, Mat{dzh + (int64_t)(2 - k) * cout, cout, 1}, Mat{W + k, (int64_t)cin * 3, 3}, dx, cin, (int)s.Lp, cin, cout, nullptr, 1.f, false,
               accumulate_dx || k > 0, (int)s.S(), 1, (s.Lp + 2) * cout, 0, 0, 0, s.Lp * cin, 0));
  return SO3X_OK;
}

// dprm (overwritten; lig_tf's slice zero: the reference never runs it) = d sum(out * dout) / d params
int backward_f32(hipStream_t st, const Dims& s, const float* prm, const float* dout, float* dprm, const Acts& a, const BwdBufs& w, const Drop& dr) {
  const POff po = param_offsets(s);
  const int64_t N = s.N(), S = s.S(), Lp = s.Lp, B = s.B, NB = B * Lp;
  const int d = s.d, H = s.H, dh = s.dh(), F = s.F, pd = s.pd(), ad = s.ad(), rd = s.rd(), pw = s.pw(), Bn = (int)B;
  hipError_t e = hipMemsetAsync(dprm, 0, (size_t)po.total * sizeof(float), st);
  if (e != hipSuccess) return (int)e;
  // head
  TRY(gemm(st, transposed(dout, 6), rowmajor(a.ha[3], d), dprm + po.wout, d, 6, d, Bn));
  TRY(colsum(st, dout, 6, B, 6, dprm + po.bout, w.part));
  float* da = w.dh0;
  float* dn = w.dh1;
  TRY(gemm(st, rowmajor(dout, 6), rowmajor(prm + po.wout, d), da, d, Bn, d, 6));
  for (int i = 2; i >= 0; i--) {   // a_{i+1} = a_i + silu(a_i Wr_i^T + br_i)
    hipLaunchKernelGGL(k_silu_bwd, dim3(nblk(B * d, 256)), dim3(256), 0, st, da, a.hz[i + 1], w.dhz, B * d);
    TRY(check_launch());
    TRY(gemm(st, transposed(w.dhz, d), rowmajor(a.ha[i], d), dprm + po.wr[i], d, d, d, Bn));
    TRY(colsum(st, w.dhz, d, B, d, dprm + po.br[i], w.part));
    TRY(gemm(st, rowmajor(w.dhz, d), rowmajor(prm + po.wr[i], d), da, d, Bn, d, d, nullptr, 1.f, false, true));   // da += dz Wr
  }
  hipLaunchKernelGGL(k_silu_bwd, dim3(nblk(B * d, 256)), dim3(256), 0, st, da, a.hz[0], w.dhz, B * d);
  TRY(check_launch());
  TRY(gemm(st, transposed(w.dhz, d), rowmajor(a.pv, pw), dprm + po.w0, pw, d, pw, Bn));
  TRY(colsum(st, w.dhz, d, B, d, dprm + po.b0, w.part));
  TRY(gemm(st, rowmajor(w.dhz, d), rowmajor(prm + po.w0, pw), w.dpv, pw, Bn, pw, d));
  (void)dn;
  // pools: pooled = Wlin xs + blin per chain kind
  for (int kind = 0; kind < 2; kind++) {
    const Pool& q = kind ? po.lig : po.rec;
    const float* dpooled = w.dpv + (kind ? 2 * d + 3 : d);
    TRY(gemm(st, transposed(dpooled, pw), rowmajor(a.xs + kind * B * d, d), dprm + q.wlin, d, d, d, Bn));
    TRY(colsum(st, dpooled, pw, B, d, dprm + q.blin, w.part));
    TRY(gemm(st, rowmajor(dpooled, pw), rowmajor(prm + q.wlin, d), w.dxs + kind * B * d, d, Bn, d, d));
  }
  hipLaunchKernelGGL(k_pool_bwd, dim3(nblk(N, 4)), dim3(256), 0, st, a.enc, a.pos, prm, po.rec, po.lig, a.wE, a.wP, a.SE, a.SP, a.xs, a.pv, w.dxs,
                     w.dpv, w.dA, w.gE, w.gP, B, Lp, d, pw);
  TRY(check_launch());
  for (int kind = 0; kind < 2; kind++) {
    const Pool& q = kind ? po.lig : po.rec;
    const float* x = a.enc + kind * NB * d;
    TRY(gemm_splitk(st, Mat{w.gE + kind * NB, 0, 1}, rowmajor(x, d), dprm + q.wpool, d, 1, d, (int)NB, w.slab, w.slab_floats));
    TRY(colsum(st, w.gE + kind * NB, 1, NB, 1, dprm + q.bpool, w.part));
    TRY(gemm_splitk(st, Mat{w.gP + kind * NB, 0, 1}, rowmajor(x, d), dprm + q.wppool, d, 1, d, (int)NB, w.slab, w.slab_floats));
    TRY(colsum(st, w.gP + kind * NB, 1, NB, 1, dprm + q.bppool, w.part));
  }
  // the encoder's final norm
  float* dcur = w.dB;    // gradient with respect to the current layer's output
  float* dalt = w.dA;
  TRY(colsum(st, w.dA, d, N, d, dprm + po.rec_tf.gF, w.part, a.h[s.T], d, a.stF));
  TRY(colsum(st, w.dA, d, N, d, dprm + po.rec_tf.bF, w.part));
  TRY(plane::ln_bwd(st, w.dA, a.h[s.T], a.stF, prm + po.rec_tf.gF, dcur, N, d));
  const float scale = 1.f / sqrtf((float)dh);
  for (int l = s.T - 1; l >= 0; l--) {
    const LayerOff lo = po.layer(po.rec_tf, l);
    const LayerActs& k = a.layer[l];
    const float* h = a.h[l];
    // norm2 over r2 = x1 + ffn(x1)
    TRY(colsum(st, dcur, d, N, d, dprm + lo.g2, w.part, k.r2, d, k.st2));
    TRY(colsum(st, dcur, d, N, d, dprm + lo.be2, w.part));
    TRY(plane::ln_bwd(st, dcur, k.r2, k.st2, prm + lo.g2, dalt, N, d));
    // dalt = d r2: flows to x1 (residual) and through [dropout] linear2 / [dropout] relu / linear1
    const float* dy2 = dalt;
    if (dr.on()) {   // the feed-forward branch sees the gradient through its output dropout (w.dO is idle until the attention block)
      TRY(dropout_apply(st, dr, l, DROP_BLOCK2, dalt, w.dO, N * d));
      dy2 = w.dO;
    }
    TRY(gemm_splitk(st, transposed(dy2, d), rowmajor(k.f, F), dprm + lo.w2, F, d, F, (int)N, w.slab, w.slab_floats));
    TRY(colsum(st, dy2, d, N, d, dprm + lo.b2, w.part));
    TRY(gemm(st, rowmajor(dy2, d), rowmajor(prm + lo.w2, F), w.dF, F, (int)N, F, d));
    TRY(plane::relu_bwd(st, w.dF, k.f, N * F, dr.on() ? dr.inv_keep() : 1.f));
    TRY(gemm_splitk(st, transposed(w.dF, F), rowmajor(k.x1, d), dprm + lo.w1, d, F, d, (int)N, w.slab, w.slab_floats));
    TRY(colsum(st, w.dF, F, N, F, dprm + lo.b1, w.part));
    TRY(plane::gemm_splitk(st, rowmajor(w.dF, F), rowmajor(prm + lo.w1, d), dalt, d, (int)N, d, F, w.slab, w.slab_floats, nullptr, true));   // dalt = d x1
    // norm1 over r1 = h + attn(h)
    TRY(colsum(st, dalt, d, N, d, dprm + lo.g1, w.part, k.r1, d, k.st1));
    TRY(colsum(st, dalt, d, N, d, dprm + lo.be1, w.part));
    TRY(plane::ln_bwd(st, dalt, k.r1, k.st1, prm + lo.g1, dcur, N, d));
    // dcur = d r1: flows to h (residual) and through [dropout] out_proj / attention / in_proj
    const float* dy1 = dcur;
    if (dr.on()) {   // (w.dF is idle from here on)
      TRY(dropout_apply(st, dr, l, DROP_BLOCK1, dcur, w.dF, N * d));
      dy1 = w.dF;
    }
    TRY(gemm_splitk(st, transposed(dy1, d), rowmajor(k.o, d), dprm + lo.wo, d, d, d, (int)N, w.slab, w.slab_floats));
    TRY(colsum(st, dy1, d, N, d, dprm + lo.bo, w.part));
    TRY(gemm(st, rowmajor(dy1, d), rowmajor(prm + lo.wo, d), w.dO, d, (int)N, d, d));
    const int64_t sq = Lp * 3 * d, sp = (int64_t)H * Lp * Lp;
    // dV = P^T dO (with dropout: the dropped-out probabilities the forward multiplied V with, rebuilt in the buffer dP takes next);
    // dP = dO V^T [through the dropout]; dS = softmax'(P, dP) / sqrt(dh) (zero at masked keys: P = 0 there); dQ = dS K; dK = dS^T Q
    const float* pv = k.probs;
    if (dr.on()) {
      TRY(dropout_apply(st, dr, l, DROP_ATTN, k.probs, w.dprobs, S * H * Lp * Lp));
      pv = w.dprobs;
    }
    TRY(gemm(st, transposed(pv, Lp), rowmajor(w.dO, d), w.dqkv + 2 * d, 3 * d, (int)Lp, dh, (int)Lp, nullptr, 1.f, false, false, (int)S, H, sp,
             Lp * Lp, Lp * d, dh, sq, dh));
    TRY(gemm(st, rowmajor(w.dO, d), transposed(k.qkv + 2 * d, 3 * d), w.dprobs, Lp, (int)Lp, (int)Lp, dh, nullptr, 1.f, false, false, (int)S, H, Lp * d,
             dh, sq, dh, sp, Lp * Lp));
    if (dr.on()) TRY(dropout_apply(st, dr, l, DROP_ATTN, w.dprobs, w.dprobs, S * H * Lp * Lp));
    TRY(plane::softmax_bwd(st, k.probs, w.dprobs, S * H * Lp, (int)Lp, scale));
    TRY(gemm(st, rowmajor(w.dprobs, Lp), rowmajor(k.qkv + d, 3 * d), w.dqkv, 3 * d, (int)Lp, dh, (int)Lp, nullptr, 1.f, false, false, (int)S, H, sp,
             Lp * Lp, sq, dh, sq, dh));
    TRY(gemm(st, transposed(w.dprobs, Lp), rowmajor(k.qkv, 3 * d), w.dqkv + d, 3 * d, (int)Lp, dh, (int)Lp, nullptr, 1.f, false, false, (int)S, H, sp,
             Lp * Lp, sq, dh, sq, dh));
    TRY(gemm_splitk(st, transposed(w.dqkv, 3 * d), rowmajor(h, d), dprm + lo.wqkv, d, 3 * d, d, (int)N, w.slab, w.slab_floats));
    TRY(colsum(st, w.dqkv, 3 * d, N, 3 * d, dprm + lo.bqkv, w.part));
    TRY(gemm(st, rowmajor(w.dqkv, 3 * d), rowmajor(prm + lo.wqkv, d), dcur, d, (int)N, d, 3 * d, nullptr, 1.f, false, true));   // dcur = d h
  }
  // dcur = d [res_emb | pos_emb | ang_emb]
  // SIREN embeddings: emb = sin(x Wp^T + bp) Wps^T + bps
  TRY(gemm_splitk(st, transposed(dcur + rd, d), rowmajor(a.snp, pd), dprm + po.wpps, pd, pd, pd, (int)N, w.slab, w.slab_floats));
  TRY(colsum(st, dcur + rd, d, N, pd, dprm + po.bpps, w.part));
  TRY(gemm(st, rowmajor(dcur + rd, d), rowmajor(prm + po.wpps, pd), w.dsp, pd, (int)N, pd, pd));
  TRY(plane::cos_mul(st, w.dsp, a.prep, N * pd));
  TRY(gemm_splitk(st, transposed(w.dsp, pd), rowmajor(a.pos, 3), dprm + po.wpp, 3, pd, 3, (int)N, w.slab, w.slab_floats));
  TRY(colsum(st, w.dsp, pd, N, pd, dprm + po.bpp, w.part));
  TRY(gemm_splitk(st, transposed(dcur + rd + pd, d), rowmajor(a.sna, ad), dprm + po.waps, ad, ad, ad, (int)N, w.slab, w.slab_floats));
  TRY(colsum(st, dcur + rd + pd, d, N, ad, dprm + po.baps, w.part));
  TRY(gemm(st, rowmajor(dcur + rd + pd, d), rowmajor(prm + po.waps, ad), w.dsa, ad, (int)N, ad, ad));
  TRY(plane::cos_mul(st, w.dsa, a.prea, N * ad));
  TRY(gemm_splitk(st, transposed(w.dsa, ad), rowmajor(a.ang, 9), dprm + po.wap, 9, ad, 9, (int)N, w.slab, w.slab_floats));
  TRY(colsum(st, w.dsa, ad, N, ad, dprm + po.bap, w.part));
  // residue convolutions, last to first; dzh's halo rows stay zero (memset once: every k_silu_bwd_halo writes rows 1 .. Lp only)
  if ((e = hipMemsetAsync(w.dzh, 0, (size_t)s.R() * d * sizeof(float), st)) != hipSuccess) return (int)e;
  for (int i = s.Cd - 1; i >= 0; i--) {
    const int cin = po.cin[i], cout = po.cout[i];
    const bool last = i == s.Cd - 1;
    if (last) {
      // (the halo layout's row width changes with cout: clear what the narrower layout will read as halo rows)
      hipLaunchKernelGGL(k_silu_bwd_halo, dim3(nblk(N * cout, 256)), dim3(256), 0, st, dcur, d, nullptr, w.dzh, a.len, S, Lp, cout);
    } else {
      if ((e = hipMemsetAsync(w.dzh, 0, (size_t)s.R() * d * sizeof(float), st)) != hipSuccess) return (int)e;
      hipLaunchKernelGGL(k_silu_bwd_halo, dim3(nblk(N * cout, 256)), dim3(256), 0, st, w.dxc, d, a.zc[i], w.dzh, a.len, S, Lp, cout);
    }
    TRY(check_launch());
    // d x_i: conv i's input gradient (+ the residual path for the ResLayers 1 .. Cd - 2: x_{i+1} = x_i + silu(z_i), w.dxc already holds d x_{i+1})
    float* dx = i == 0 ? nullptr : w.dxc;
    const bool residual = i >= 1 && i <= s.Cd - 2;
    TRY(conv_bwd(st, s, a.xc[i], cin, prm + po.cw[i], cout, w.dzh, dprm + po.cw[i], dprm + po.cb[i], dx, residual, w));
  }
  return SO3X_OK;
}

}  // namespace prot
}  // namespace so3x

// ------------------------------------------------------------------------------------------------ C ABI
using namespace so3x::prot;

namespace so3x { namespace prot {
// so3x_protnet_bf16.hip
bool bf16_supported(const Dims& s);
size_t bf16_workspace_bytes(const Dims& s, int64_t n_rec, int64_t n_lig);
int forward_bf16(hipStream_t st, const Dims& s, const float* prm, const float* rres, const float* rpos, const float* rang, const int64_t* roff,
                 int64_t n_rec, const float* lres, const float* lpos, const float* lang, const int64_t* loff, int64_t n_lig, const int64_t* t,
                 float* out, float* pool_out, float* enc_out, void* workspace);
} }

extern "C" {

int64_t so3x_protnet_param_count(int dim, int heads, int t_depth, int c_depth) {
  Dims s{0, 1, dim, heads, t_depth, c_depth, 2048};
  if (!dims_ok(s)) return SO3X_ERR_INVALID_ARG;
  return param_offsets(s).total;
}

size_t so3x_protnet_stash_bytes(int64_t B, int64_t max_len, int dim, int heads, int t_depth, int c_depth, int precision) {
  Dims s{B, max_len, dim, heads, t_depth, c_depth, 2048};
  if (!dims_ok(s) || precision != SO3X_PREC_F32) return 0;
  return carve_acts(s, nullptr, true).bytes;
}

size_t so3x_protnet_workspace_bytes(int64_t B, int64_t max_len, int64_t n_rec, int64_t n_lig, int dim, int heads, int t_depth, int c_depth,
                                    int precision) {
  Dims s{B, max_len, dim, heads, t_depth, c_depth, 2048};
  if (!dims_ok(s)) return 0;
  if (precision == SO3X_PREC_BF16) return bf16_supported(s) ? bf16_workspace_bytes(s, n_rec, n_lig) : 0;
  const size_t f = carve_acts(s, nullptr, false).bytes, b = carve_bwd(s, nullptr).bytes;
  return f > b ? f : b;
}

int so3x_protnet_fwd(so3x_stream_t st, const float* params, const float* rec_res, const float* rec_pos, const float* rec_ang, const int64_t* rec_off,
                     int64_t n_rec, const float* lig_res, const float* lig_pos, const float* lig_ang, const int64_t* lig_off, int64_t n_lig,
                     const int64_t* t, float* out, float* pool_out, float* enc_out, int64_t B, int64_t max_len, int dim, int heads, int t_depth,
                     int c_depth, int precision, void* stash, void* workspace, size_t workspace_bytes, float dropout_p, uint64_t seed,
                     uint64_t rng_offset) {
  Dims s{B, max_len, dim, heads, t_depth, c_depth, 2048};
  if (!dims_ok(s) || n_rec < 0 || n_lig < 0) return SO3X_ERR_INVALID_ARG;
  if (!(dropout_p >= 0.f && dropout_p < 1.f) || (dropout_p > 0.f && (!stash || precision != SO3X_PREC_F32))) return SO3X_ERR_INVALID_ARG;   // dropout = a training forward
  if (B && (!params || !rec_res || !rec_pos || !rec_ang || !rec_off || !lig_res || !lig_pos || !lig_ang || !lig_off || !t || !out)) return SO3X_ERR_INVALID_ARG;
  if (precision != SO3X_PREC_F32 && precision != SO3X_PREC_BF16) return SO3X_ERR_UNSUPPORTED;
  if (precision == SO3X_PREC_BF16 && (!bf16_supported(s) || stash)) return SO3X_ERR_UNSUPPORTED;   // (the bf16 form is the inference form)
  if (B == 0) return SO3X_OK;
  if (!workspace || workspace_bytes < so3x_protnet_workspace_bytes(B, max_len, n_rec, n_lig, dim, heads, t_depth, c_depth, precision)) return SO3X_ERR_WORKSPACE;
  if (precision == SO3X_PREC_BF16)
    return forward_bf16((hipStream_t)st, s, params, rec_res, rec_pos, rec_ang, rec_off, n_rec, lig_res, lig_pos, lig_ang, lig_off, n_lig, t, out, pool_out, enc_out, workspace);
  const Acts a = stash ? carve_acts(s, stash, true) : carve_acts(s, workspace, false);
  const Drop dr{dropout_p, seed, rng_offset};
  // (with a stash the workspace is idle in the forward: the backward's dP buffer holds a layer's dropped-out probabilities)
  return forward_f32((hipStream_t)st, s, params, rec_res, rec_pos, rec_ang, rec_off, lig_res, lig_pos, lig_ang, lig_off, t, out, pool_out, enc_out, a, dr,
                     dr.on() ? carve_bwd(s, workspace).dprobs : nullptr);
}

int so3x_protnet_bwd(so3x_stream_t st, const float* params, const float* dout, float* dparams, int64_t B, int64_t max_len, int dim, int heads,
                     int t_depth, int c_depth, int precision, const void* stash, void* workspace, size_t workspace_bytes, float dropout_p,
                     uint64_t seed, uint64_t rng_offset) {
  Dims s{B, max_len, dim, heads, t_depth, c_depth, 2048};
  if (!dims_ok(s) || !dparams || (B && (!params || !dout || !stash)) || !(dropout_p >= 0.f && dropout_p < 1.f)) return SO3X_ERR_INVALID_ARG;
  if (precision != SO3X_PREC_F32) return SO3X_ERR_UNSUPPORTED;
  if (B == 0) {
    hipError_t e = hipMemsetAsync(dparams, 0, (size_t)param_offsets(s).total * sizeof(float), (hipStream_t)st);
    return e == hipSuccess ? SO3X_OK : (int)e;
  }
  if (!workspace || workspace_bytes < so3x_protnet_workspace_bytes(B, max_len, 0, 0, dim, heads, t_depth, c_depth, precision)) return SO3X_ERR_WORKSPACE;
  const Acts a = carve_acts(s, const_cast<void*>(stash), true);
  const BwdBufs w = carve_bwd(s, workspace);
  return backward_f32((hipStream_t)st, s, params, dout, dparams, a, w, Drop{dropout_p, seed, rng_offset});
}

}  // extern "C"
