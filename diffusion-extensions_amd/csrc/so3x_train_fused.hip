// so3x_train_fused.hip -- one training step of SO3Diffusion(RotPredict(65, "skewvec")) under loss_type="skewvec"
// (reference so3_train.py:73-75: loss = process(truepos); loss.backward()) as ONE kernel: noise draw + q_sample + regression
// target (diffusion.py:339-355, distributions.py:33-51), the network forward (so3_train.py:39-49), the MSE and its gradient
// (diffusion.py:357) and the whole backward down to per-workgroup dW partial slabs.  Nothing per-sample goes through HBM:
// a sample costs its 36 bytes of x_0 in (the staged step -- so3x_mlp_bwd.hip -- wrote and re-read ~1.4 KB per sample:
// x_t, target, timestep, dout and a 544-byte pre-activation stash between three kernels).
//
// Nothing in the math needs a grid-wide barrier between forward and backward: d loss / d out = 2 (out - target) / (3 n) is
// per sample with n known up front; only the REPORTED loss is a reduction (arrival ticket, as in the staged forward).
//
// Workgroup = 8 waves, one per CU (LDS 158 KB), as k_bwd_fused:
//   * waves 0-3 ("chain"): one 32-sample tile per round each -- forward through the LDS-resident weight image with the
//     table SiLU, whose 16-byte entries (GATHER_TD image, so3x_mlp.hpp) yield silu(z) AND silu'(z) in one lookup: the
//     activations are kept as packed bf16 pairs (they ARE the next layer's MFMA operand and, later, the H_l image of the dW
//     products) and the derivatives as packed f16 pairs, 4 x (17 + 17) registers -- the pre-activations themselves are never
//     kept, so the backward has no transcendental and no stash to wait for: dZ_{l-1} = dH_l * silu'(Z_{l-1}) is one
//     multiply per value.  Then the dZ chain on the transposed image, handing dZ_l / H_l to the dW waves through the per-wave
//     LDS images of so3x_train.hpp, two barriers per layer.
//   * waves 4-7 ("dW"): the 39 dW tiles as persistent MFMA accumulators (dw_row), and -- in the time the chain waves spend in
//     their forward, when the images are idle -- the NOISING of the tiles two rounds ahead: lane = sample (Philox block,
//     timestep, inverse-CDF angle, Rodrigues, q_sample, target: k_q_sample_target's arithmetic, bit for bit), 64 samples per
//     pass every other round; x_t leaves as the packed bf16 pairs the network's layer 0 and the H_0 image want, with the
//     target and the timestep, through a 4.5 KB hand-over buffer in LDS.
#include "so3x_common.hpp"
#include "so3x_igso3.hpp"
#include "so3x_mlp.hpp"
#include "so3x_reverse_step.hpp"
#include "so3x_train.hpp"

using namespace so3x;
using namespace so3x::mlp;
using namespace so3x::train;

namespace {

constexpr int PREC = SO3X_PREC_BF16, VAR = GATHER_TD, FB = frag_bytes<PREC>();
constexpr int IMG = image_bytes<PREC, VAR>();                 // 53 fragments + the 4 KB (silu, silu') table
constexpr int REC_DW = 8;                                     // hand-over record of a sample: x_t as 5 bf16 pairs, target[3]
constexpr int HAND_BYTES = 128 * REC_DW * 4 + 128 * 4;        // ... + the timesteps
constexpr int WTB = wt_bytes<PREC>();                         // transposed image: 48 fragments
// LDS: the hand-over images FIRST -- their 150 distinct read / store addresses per round are (per-lane base) + constant, and a
// DS instruction's offset field holds 16 bits: behind the 107 KB of weight images every one of them cost an address register
// (60 VGPRs in the dW waves, beside 160 accumulators) -- then the hand-over records, the loss scratch and the hand-shake words,
// the two weight images
constexpr int LDS_FIMG = 0, LDS_HAND = LDS_FIMG + 4 * FIMG_BYTES, LDS_RED = LDS_HAND + HAND_BYTES, LDS_IMG = LDS_RED + 256;
constexpr int LDS_WT = LDS_IMG + IMG, LDS_TOTAL = LDS_WT + WTB;
static_assert(LDS_TOTAL <= 160 * 1024 && LDS_HAND % 16 == 0 && LDS_IMG % 16 == 0, "one workgroup per CU");

__device__ __forceinline__ uint32_t pack_f16x2(float a, float b) {
  typedef _Float16 h2 __attribute__((ext_vector_type(2)));
  return __builtin_bit_cast(uint32_t, h2{(_Float16)a, (_Float16)b});
}
__device__ __forceinline__ float f16_lo(uint32_t w) {
  typedef _Float16 h2 __attribute__((ext_vector_type(2)));
  return (float)__builtin_bit_cast(h2, w)[0];
}
__device__ __forceinline__ float f16_hi(uint32_t w) {
  typedef _Float16 h2 __attribute__((ext_vector_type(2)));
  return (float)__builtin_bit_cast(h2, w)[1];
}

// what a noising lane hands to the chain wave that owns its sample
struct Hand { uint32_t xb[5]; float tg[3]; int tt; };

// One sample of SO3Diffusion.forward's front half (diffusion.py:339-355, 373): k_q_sample_target's arithmetic (so3x_diffusion.hip),
// operation for operation -- the staged step and this one draw the same noise, timesteps and targets bit for bit.
// EXPLICIT: the caller supplies the sampler's draws (axes, unif: the parity paths); otherwise they come from the sample's Philox block.
// sc = the sample index clamped into [0, n) (dead lanes compute on it and are overwritten afterwards).
template <bool EXPLICIT>
__device__ __forceinline__ Hand noise_sample(const NoiseArgs& na, uint64_t rng_offset, int64_t wrow_t, int64_t sc, bool live,
                                             float* __restrict__ x_t_out) {
  const int T = na.T;
  auto drawn_t = [&](uint32_t w) -> int64_t { return (int64_t)(((uint64_t)w * (uint64_t)T) >> 32); };
  auto clamp_t = [&](int64_t v) -> int64_t { return v < 0 ? 0 : (v >= T ? T - 1 : v); };
  Philox4 r{0u, 0u, 0u, 0u};
  if (!na.t || !EXPLICIT) r = philox4x32_10(na.seed, (uint64_t)(na.index_base + sc), rng_offset);
  const int64_t tt = na.t ? clamp_t(na.t[sc]) : drawn_t(r.w);
  if (na.t_draw && live) na.t_draw[sc] = tt;
  float ax[3], u;
  if constexpr (EXPLICIT) {
    const float a0 = na.axes[sc * 3], a1 = na.axes[sc * 3 + 1], a2 = na.axes[sc * 3 + 2];
    const float nrm = sqrtf(a0 * a0 + a1 * a1 + a2 * a2);             // distributions.py:36
    ax[0] = a0 / nrm; ax[1] = a1 / nrm; ax[2] = a2 / nrm;
    const float n2 = sqrtf(ax[0] * ax[0] + ax[1] * ax[1] + ax[2] * ax[2]);  // util.py:201
    ax[0] /= n2; ax[1] /= n2; ax[2] /= n2;
    u = na.unif[sc];
  } else {
    unit_axis(r.x, r.y, ax);
    u = u01(r.z);
  }
  const float* row = na.trap_q + tt * 999;
  const float* wrow = wrow_t >= 0 ? na.trap_q + wrow_t * 999 : row;
  const float ang = igso3_angle_global(row, wrow, SO3X_KNOTS_DATA, u, na.guide_q ? na.guide_q + tt * kGuidePitch : nullptr);
  float nz[9], x[9], w[3], xs[9], xt[9], lw[3];
  exp_axis_angle(ax, ang, nz);
  load_rot9(na.x0, sc, x);
  const float k = na.sched[S_SQRT_AC * T + tt];
  log3(x, w);
  w[0] *= k; w[1] *= k; w[2] *= k;
  exp3(w, xs);                      // so3_scale(x_start, sqrt(abar_t)), diffusion.py:344-345
  mul33(xs, nz, xt);                // x_blend @ noise, :346
  if (x_t_out && live) store_rot9(x_t_out, sc, xt);
  // skew2vec(log_rmat(noise)) / eps, :355 -- of a noise built here from a unit axis and an angle in [0, pi] it IS axis * angle
  if constexpr (EXPLICIT) log3(nz, lw);
  else { lw[0] = ax[0] * ang; lw[1] = ax[1] * ang; lw[2] = ax[2] * ang; }
  const float ie = 1.0f / na.sched[S_SQRT_1MAC * T + tt];
  Hand hd;
  hd.tg[0] = lw[0] * ie; hd.tg[1] = lw[1] * ie; hd.tg[2] = lw[2] * ie;
  hd.xb[0] = pack_bf16x2(xt[0], xt[1]); hd.xb[1] = pack_bf16x2(xt[2], xt[3]); hd.xb[2] = pack_bf16x2(xt[4], xt[5]);
  hd.xb[3] = pack_bf16x2(xt[6], xt[7]); hd.xb[4] = pack_bf16x2(xt[8], 1.0f);
  hd.tt = (int)tt;
  if (!live) {  // a column past the batch: well-defined finite inputs (its dZ is an exact zero in every layer)
    hd.xb[0] = 0x00003F80u; hd.xb[1] = 0u; hd.xb[2] = 0x00003F80u; hd.xb[3] = 0u; hd.xb[4] = 0x3F803F80u;
    hd.tg[0] = hd.tg[1] = hd.tg[2] = 0.0f;
    hd.tt = 0;
  }
  return hd;
}

// Index of the 256-entry SiLU table for a pre-activation coordinate: round to nearest, saturated to 0..255.
__device__ __forceinline__ unsigned tab_index(float u) {
  unsigned idx = __builtin_amdgcn_cvt_pk_u8_f32(u, 0u, 0u);
#if defined(SO3X_AB_BUILD) && defined(FUSED_AB_TAB_SPREAD)
  // counters-only build (tools/ab/fused_bounds.sh): lane l reads an entry of bank group l mod 16, so the sixteen 16-byte reads of
  // one LDS clock never share a bank; the values are wrong, the instruction stream is the same
  idx = (idx & ~15u) | (threadIdx.x & 15u);
#endif
  return idx;
}

// SiLU and its derivative for one layer's 33 pre-activation coordinates u = 16 z + 127.5 (the MFMAs emit them: GATHER_TD image):
// hp = silu(z) as packed bf16 pairs in the order of the next layer's B operand (word w of k-step k = hp[4 k + w]; word 16 = the
// fifth k-step's first: feature 64 | the constant ones of rows 68, 69), dp = silu'(z) as packed f16 pairs in the same order.
// The lookups of eight values are issued together and consumed afterwards (one LDS round trip per group).
__device__ __forceinline__ void activate_td(const f32x16 (&acc)[3], uint32_t (&hp)[17], uint32_t (&dp)[17], int h, const char* tab) {
  // Eight groups of four values + the single value of tile 2 as a two-deep software pipeline: the lookups of group g + 1 are issued in
  // front of the arithmetic of group g, so a lone wave (nothing else on its SIMD hides an LDS round trip) pays the latency once per
  // layer instead of once per group (five sequential groups of eight cost 1.4 k cycles per layer: phase stamps, round 4)
  constexpr int G = 4, NG = 8;
  float4 e[2][G];
  float u[2][G];
  auto lookups = [&](int g) {
#pragma unroll
    for (int i = 0; i < G; i++) {
      const int q = G * g + i;
      u[g & 1][i] = acc[q >> 4][q & 15];
      const unsigned idx = tab_index(u[g & 1][i]);
      e[g & 1][i] = *reinterpret_cast<const float4*>(tab + idx * 16);
    }
  };
  const float ul = acc[2][0];
  lookups(0);
  float4 el;
#pragma unroll
  for (int g = 0; g < NG; g++) {
    if (g + 1 < NG) lookups(g + 1);
    else el = *reinterpret_cast<const float4*>(tab + tab_index(ul) * 16);
#pragma unroll
    for (int i = 0; i < G; i += 2) {
      const float4 e0 = e[g & 1][i], e1 = e[g & 1][i + 1];
      const float u0 = u[g & 1][i], u1 = u[g & 1][i + 1];
      const int w = (G * g + i) >> 1;
      hp[w] = pack_bf16x2(fmaf(e0.y, u0, e0.x), fmaf(e1.y, u1, e1.x));
      dp[w] = pack_f16x2(fmaf(e0.w, u0, e0.z), fmaf(e1.w, u1, e1.z));
      // the packed words ARE the parked state: opaque, so that the compiler keeps them and not their two fp32 sources each
      // (it sank the packing to the backward's uses and spilled 264 fp32 values per round)
      asm volatile("" : "+v"(hp[w]), "+v"(dp[w]));
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  hp[16] = h ? 0x3F803F80u : pack_bf16x2(fmaf(el.y, ul, el.x), 0.0f);   // rows 68, 69: the constant ones (bias / table offset carriers)
  dp[16] = h ? 0u : pack_f16x2(fmaf(el.w, ul, el.z), 0.0f);
  asm volatile("" : "+v"(hp[16]), "+v"(dp[16]));
}

__device__ __forceinline__ void operand_of(const uint32_t (&hp)[17], Tile<PREC>& t) {
  typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
#pragma unroll
  for (int k = 0; k < 4; k++) t.b[k] = __builtin_bit_cast(bf16x8, u32x4_t{hp[4 * k], hp[4 * k + 1], hp[4 * k + 2], hp[4 * k + 3]});
  t.b[4] = __builtin_bit_cast(bf16x8, u32x4_t{hp[16], 0u, 0u, 0u});
}

// One hidden layer of the forward, MFMAs and activation as ONE stream: output tile 0's five MFMAs, then its sixteen activations
// with the ten MFMAs of tiles 1 and 2 issued INTO the LDS round trips of its lookup groups (a lone wave waits ~100 cycles per group
// there, and a 32-cycle MFMA issue slot is as good a use of the wait as any), then tiles 1 and 2's activations.  hidden_layer +
// activate_td one after the other left the matrix pipe idle through 1.3 k cycles of activation and the wave idle through the
// MFMAs' 0.5 k.
__device__ __forceinline__ void hidden_fwd_td(const char* __restrict__ wl, const Tile<PREC>& in, f32x16 (&acc)[3], uint32_t (&hp)[17],
                                              uint32_t (&dp)[17], int h, const char* tab, int lane) {
  const bf16x8* w = reinterpret_cast<const bf16x8*>(wl);
  constexpr int G = 4, NG = 8;
  __builtin_amdgcn_sched_barrier(0);
  {
    f32x16 a = zero16<PREC>();
#pragma unroll
    for (int ks = 0; ks < 5; ks++) a = mfma_bf16(w[ks * 64 + lane], in.b[ks], a);
    acc[0] = a;
  }
  acc[1] = zero16<PREC>();
  acc[2] = zero16<PREC>();
  float4 e[2][G];
  float u[2][G];
  auto lookups = [&](int g) {
#pragma unroll
    for (int i = 0; i < G; i++) {
      const int q = G * g + i;
      u[g & 1][i] = acc[q >> 4][q & 15];
      const unsigned idx = tab_index(u[g & 1][i]);
      e[g & 1][i] = *reinterpret_cast<const float4*>(tab + idx * 16);
    }
  };
  auto mm = [&](int m) {  // MFMA m of tiles 1, 2: (to, ks) = (1 + m / 5, m % 5)
    const int to = 1 + m / 5, ks = m % 5;
    acc[to] = mfma_bf16(w[((to * 5) + ks) * 64 + lane], in.b[ks], acc[to]);
  };
  __builtin_amdgcn_sched_barrier(0);
  lookups(0);
  float4 el;
  float ul = 0.0f;
#pragma unroll
  for (int g = 0; g < NG; g++) {
    // tile 1 has to be complete in front of group 3's look-ahead (its values are group 4's), tile 2 in front of the last value
    if (g == 0) { mm(0); mm(1); mm(2); }
    if (g == 1) { mm(3); mm(4); mm(5); }
    if (g == 2) { mm(6); mm(7); }
    if (g == 3) { mm(8); mm(9); }
    if (g + 1 < NG) lookups(g + 1);
    else { ul = acc[2][0]; el = *reinterpret_cast<const float4*>(tab + tab_index(ul) * 16); }
#pragma unroll
    for (int i = 0; i < G; i += 2) {
      const float4 e0 = e[g & 1][i], e1 = e[g & 1][i + 1];
      const float u0 = u[g & 1][i], u1 = u[g & 1][i + 1];
      const int wd = (G * g + i) >> 1;
      hp[wd] = pack_bf16x2(fmaf(e0.y, u0, e0.x), fmaf(e1.y, u1, e1.x));
      dp[wd] = pack_f16x2(fmaf(e0.w, u0, e0.z), fmaf(e1.w, u1, e1.z));
      asm volatile("" : "+v"(hp[wd]), "+v"(dp[wd]));   // (the packed words are the parked state: see activate_td)
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  hp[16] = h ? 0x3F803F80u : pack_bf16x2(fmaf(el.y, ul, el.x), 0.0f);
  dp[16] = h ? 0u : pack_f16x2(fmaf(el.w, ul, el.z), 0.0f);
  asm volatile("" : "+v"(hp[16]), "+v"(dp[16]));
}

// ---- the dW waves: k_bwd_fused's dW role (so3x_mlp_bwd.hip) + the noising of the tiles ahead -----------------------------------
struct Geo { int64_t n, ntiles, nchain, rounds; };
// -DTF_STAMPS (timing build, tools/ab/fused_stamps.py; results of `out` destroyed): workgroup 0's chain wave 0 and dW wave 0 leave
// s_memtime stamps of a round's phases in the `out` buffer: stamps[role][round][32]
#ifdef TF_STAMPS
#define TF_STAMP(role, k) do { if (blockIdx.x == 0 && stamp_base && rd < 64) { const uint64_t now_ = __builtin_amdgcn_s_memtime(); \
    if (lane == 0) stamp_base[((role) * 64 + rd) * 32 + (k)] = now_; } } while (0)
#else
#define TF_STAMP(role, k) do { } while (0)
#endif
constexpr int RING = 3;  // operand ring of the dW products: stages (image, k-step) in flight, 16 registers each

// ---- hand-shakes per IMAGE instead of workgroup barriers (round 4, second half) -------------------------------------------------
// With two s_barriers per layer the chain waves stored a layer's images, THEN the dW waves took their 64 transposed reads per
// layer, and neither did the one while the other happened.  Now a chain wave announces the n-th filling of ITS image as
// ready[w] = n + 1 (n = 5 round + layer index); the dW waves walk the four waves' images in order, wait for each one's filling
// and count themselves out of it behind their last read (done[w] += 1); the chain wave refills its image once done[w] = 4 n.
// A chain wave whose image the dW waves have left goes on while they are still in its siblings'; the four fall into a
// stagger.  handed[w] = the round whose records the wave's dW partner has put into the hand-over buffer.  One wave's LDS
// operations are processed in issue order, so a flag written behind the data is seen behind the data; the accesses are inline
// assembly with a memory clobber so that the compiler keeps that order too.  Every wait gives up after ~0.1 s and the
// step then reports a NaN loss (a loud wrong result instead of a hung GPU).
// (Tried and dropped, profiles/r04_ab_train_fused_sync.json: TWO images per chain wave, with the transposed weight image left in
//  global memory to make room -- the chain waves ran a layer ahead, and lost more to the 15 L2 fetches per layer and wave than
//  the decoupling gave: 156 us against 146.)
constexpr int LDS_READY = LDS_RED + 80, LDS_DONE = LDS_RED + 96, LDS_HANDED = LDS_RED + 112;   // [4], [4], [4] words
static_assert(LDS_HANDED + 16 <= LDS_IMG, "the flags live behind the loss scratch");
__device__ __forceinline__ uint32_t img_seq(int64_t rd, int k) { return 5u * (uint32_t)rd + (uint32_t)k; }  // fillings in front of (rd, k)
__device__ __forceinline__ uint32_t lds_peek(uint32_t addr) {
  uint32_t v;
  asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
  return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
}
// `gave_up` (wave-uniform, a scalar register): set when the wait ran out of patience after ~0.1 s -- the wave goes on (the words
// carry absolute counts: one lost hand-shake does not take the later ones with it) and the step's loss comes out NaN.
__device__ __forceinline__ void lds_wait_ge(uint32_t addr, uint32_t target, uint32_t& gave_up) {
  int spins = 0;
  while ((int32_t)(lds_peek(addr) - target) < 0) {
    __builtin_amdgcn_s_sleep(1);
    if (++spins > (1 << 21)) { gave_up = 1u; break; }
  }
}
// The same wait with the word READ EARLY: `seen` was loaded (lds_early) some hundred cycles ahead, as an ordinary LDS load in
// the wave's stream -- looking at it costs no drain of the wave's LDS queue (lds_peek's s_waitcnt lgkmcnt(0) does: in the dW
// waves that emptied the operand ring in front of every image, ~200 cycles four times a layer).  A stale "not yet" falls back
// to the polling loop; nothing that follows may be moved in front of the check (the empty asm is the compiler's fence).
__device__ __forceinline__ uint32_t lds_early(const char* lds, uint32_t addr) {
  return *reinterpret_cast<const volatile uint32_t*>(lds + addr);
}
__device__ __forceinline__ void lds_wait_ge_seen(uint32_t seen, uint32_t addr, uint32_t target, uint32_t& gave_up) {
  if ((int32_t)((uint32_t)__builtin_amdgcn_readfirstlane((int)seen) - target) < 0) lds_wait_ge(addr, target, gave_up);
  asm volatile("" ::: "memory");
}
__device__ __forceinline__ void lds_post(uint32_t addr, uint32_t val) {   // every lane writes the same word
  asm volatile("ds_write_b32 %0, %1" :: "v"(addr), "v"(val) : "memory");
}
__device__ __forceinline__ void lds_count(uint32_t addr, int lane) {      // + 1, once per wave
  if (lane == 0) asm volatile("ds_add_u32 %0, %1" :: "v"(addr), "v"(1u) : "memory");
}

template <int DWI, bool EXPLICIT>
__device__ __forceinline__ void dw_role_fused(char* lds, const Geo& g, const NoiseArgs& na, uint64_t rng_offset, int64_t wrow_t,
                                              float* __restrict__ x_t_out, float* __restrict__ slabs, int lane, uint32_t& gave_up,
                                              uint64_t* stamp_base = nullptr) {
  const char* fimg_all = lds + LDS_FIMG;
  const int col = lane & 31, h = lane >> 5;
  f32x16 acc[10];  // [3 slot + ti] for the hidden layers, [9] = the wave's tile of the output layer
#pragma unroll
  for (int k = 0; k < 10; k++) acc[k] = zero16<PREC>();
  // a noising pass covers the wave's tile of round `ra` (lanes 0..31) and of round ra + 1 (lanes 32..63)
  auto pass = [&](int64_t ra) -> Hand {
    // (the lane index is made opaque here: nothing per-lane of the pass -- sample index, row and table addresses -- is computed
    //  in front of the round loop and kept alive, or spilled, beside the 160 accumulator registers)
    int lane_o = lane;
    asm volatile("" : "+v"(lane_o));
    const int col = lane_o & 31, h = lane_o >> 5;
    const int64_t r = ra + h;
    const int64_t tile = r * g.nchain + (int64_t)blockIdx.x * 4 + DWI, s = tile * 32 + col;
    const bool live = r < g.rounds && tile < g.ntiles && s < g.n;
    return noise_sample<EXPLICIT>(na, rng_offset, wrow_t, live ? s : g.n - 1, live, x_t_out);
  };
  auto hand_over = [&](const Hand& hd) {
    int lane_o = lane;
    asm volatile("" : "+v"(lane_o));  // (the two record addresses are made here, not kept -- or spilled -- across the round loop)
    uint32_t* rec = reinterpret_cast<uint32_t*>(lds + LDS_HAND) + (DWI * 32 + (lane_o & 31)) * REC_DW;
    int* ht = reinterpret_cast<int*>(lds + LDS_HAND + 128 * REC_DW * 4) + DWI * 32 + (lane_o & 31);
    *reinterpret_cast<uint4*>(rec) = uint4{hd.xb[0], hd.xb[1], hd.xb[2], hd.xb[3]};
    *reinterpret_cast<uint4*>(rec + 4) = uint4{hd.xb[4], __float_as_uint(hd.tg[0]), __float_as_uint(hd.tg[1]), __float_as_uint(hd.tg[2])};
    *ht = hd.tt;
  };
  Hand hd = pass(0);
  if (h == 0) hand_over(hd);
  __syncthreads();  // P: round 0's samples are in the hand-over buffer (the last workgroup barrier before the loss)
  FimgReadLane RL = fimg_read_lane(lane);
  for (int64_t rd = 0; rd < g.rounds; rd++) {
    asm volatile("" : "+v"(RL.off[0][0]), "+v"(RL.off[0][1]), "+v"(RL.off[1][0]), "+v"(RL.off[1][1]));
    if (DWI == 0) TF_STAMP(1, 0);
#pragma unroll
    for (int l = 4; l >= 0; l--) {
      const uint32_t v = img_seq(rd, 4 - l) + 1u;   // the filling of the images this layer's products read
      if (l == 4) {
        // the partner chain wave has read round rd's records once its forward is through, i.e. once its layer-4 image is out:
        // then round rd + 1's go in, and the partner may take them in front of its layer-0 stores
        lds_wait_ge(LDS_READY + 4 * DWI, v, gave_up);
        if (h == (int)((rd & 1) ^ 1)) hand_over(hd);
        lds_post(LDS_HANDED + 4 * DWI, (uint32_t)rd + 1u);
      }
      if (DWI == 0) TF_STAMP(1, 2 + 3 * (4 - l));
      // The layer's products as EIGHT stages (image w, k-step ks) through a RING-slot operand ring: the transposed reads of
      // stage s + RING - 1 are issued in front of the MFMAs of stage s (left to itself the compiler reused two operand registers
      // for every MFMA: an LDS round trip per MFMA).  An image's first stage waits for its version, its last counts the wave out.
      if (l == 4 ? DWI < 3 : dw_row(DWI, l) != 3) {
        const int to = l == 4 ? 0 : dw_row(DWI, l), sl = l == 4 ? 3 : dw_slot(DWI, l);
        constexpr int NB = 3;
        bf16x8 ra[RING], rb[RING][NB];
        uint32_t seen = lds_early(lds, LDS_READY);   // image 0's word; image w + 1's is read behind image w's first stage
        auto load = [&](int st) {
          if ((st & 1) == 0) lds_wait_ge_seen(seen, LDS_READY + 4 * (st >> 1), v, gave_up);
          const char* im = fimg_all + (st >> 1) * FIMG_BYTES;
          ra[st % RING] = fimg_frag(im, RL, 32 * to, st & 1);
          if (l == 4) rb[st % RING][0] = fimg_frag(im, RL, 96 + 32 * DWI, st & 1);
          else {
#pragma unroll
            for (int ti = 0; ti < 3; ti++) rb[st % RING][ti] = fimg_frag(im, RL, 96 + 32 * ti, st & 1);
          }
          if (st & 1) lds_count(LDS_DONE + 4 * (st >> 1), lane);
          else if (st < 6) seen = lds_early(lds, LDS_READY + 4 * ((st >> 1) + 1));
        };
#pragma unroll
        for (int st = 0; st < RING - 1; st++) load(st);
#pragma unroll
        for (int st = 0; st < 8; st++) {
          if (st + RING - 1 < 8) load(st + RING - 1);
          __builtin_amdgcn_sched_barrier(0);
          if (l == 4) acc[9] = mfma_bf16(ra[st % RING], rb[st % RING][0], acc[9]);
          else {
#pragma unroll
            for (int ti = 0; ti < 3; ti++) acc[3 * sl + ti] = mfma_bf16(ra[st % RING], rb[st % RING][ti], acc[3 * sl + ti]);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      } else {  // the wave without a row in this layer: it still counts itself out of every image (once the image exists)
#pragma unroll
        for (int w = 0; w < 4; w++) {
          lds_wait_ge(LDS_READY + 4 * w, v, gave_up);
          lds_count(LDS_DONE + 4 * w, lane);
        }
      }
      if (DWI == 0) TF_STAMP(1, 3 + 3 * (4 - l));
      // behind an even round's last products: the samples of rounds rd + 2, rd + 3 are drawn while the chain waves run the NEXT
      // round's forward (the images are idle until its end)
      if (l == 0 && !(rd & 1)) hd = pass(rd + 2);
      if (DWI == 0 && l == 0) TF_STAMP(1, 1);
      if (DWI == 0) TF_STAMP(1, 4 + 3 * (4 - l));
    }
  }
#ifdef TF_STAMPS
  if (DWI == 0 && blockIdx.x == 0 && stamp_base && lane == 0) stamp_base[(64 + 63) * 32 + 4] = __builtin_amdgcn_s_memtime();
#endif
  // ---- slab: D-layout lane column = H feature (in), register rows = dZ feature (out)
  float* slab = slabs + (size_t)blockIdx.x * NPARAMS_MAX;
  int colw = col, hw = h;
  asm volatile("" : "+v"(colw), "+v"(hw));  // (no slab addresses computed -- and spilled -- in front of the round loop)
  auto write_tile = [&](const f32x16& a, int l, int to, int ti) {
    const int in_f = 32 * ti + colw;
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const int out = l < 4 ? 32 * to + row_of(r, hw) : head_of_row(row_of(r, hw));
      if (out < 0 || out >= (l < 4 ? D : 3)) continue;
      int pc;  // weight column, -2 = bias, -1 = padding
      if (l == 0) pc = in_f < 9 ? in_f : (in_f == 9 ? -2 : (in_f < 66 ? in_f - 1 : -1));
      else pc = in_f < D ? in_f : (in_f == ONE_ROW ? -2 : -1);
      if (pc == -1) continue;
      const int base = l * LAYER_STRIDE;
      slab[pc >= 0 ? base + out * D + pc : base + (l < 4 ? D : 3) * D + out] = a[r];
    }
  };
#pragma unroll
  for (int l = 0; l < 4; l++) {
    if (dw_row(DWI, l) == 3) continue;
#pragma unroll
    for (int ti = 0; ti < 3; ti++) write_tile(acc[3 * dw_slot(DWI, l) + ti], l, dw_row(DWI, l), ti);
  }
  if (DWI < 3) write_tile(acc[9], 4, 0, DWI);
#ifdef TF_STAMPS
  if (DWI == 0 && blockIdx.x == 0 && stamp_base && lane == 0) stamp_base[(64 + 63) * 32 + 5] = __builtin_amdgcn_s_memtime();
#endif
}

template <bool EXPLICIT>
__global__ void __launch_bounds__(512, 2)
k_train_fused(const void* __restrict__ gimg, const void* __restrict__ gwt, const float* __restrict__ beff_tab,
              const uint4* __restrict__ h0_tab, NoiseArgs na, float* __restrict__ x_t_out, float* __restrict__ out,
              float* __restrict__ slabs, int64_t n, LossArgs la) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
#ifdef TF_STAMPS
  const uint64_t t_entry = __builtin_amdgcn_s_memtime(), rt_entry = __builtin_amdgcn_s_memrealtime();
#endif
  char* img_lds = lds + LDS_IMG;
  char* fimg_all = lds + LDS_FIMG;
  const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), col = lane & 31, h = lane >> 5;
  // The chain waves bring the weight images into LDS and clear the hand-over images and the hand-shake words WHILE the dW waves
  // draw round 0's and 1's samples (neither needs the other's result before the one barrier P below; 11 us of prologue with the
  // two one after the other)
  if (wid < 4) {
    auto copy = [&](const void* src, char* dst, int bytes) {
      for (int i = threadIdx.x; i < bytes / 16; i += 256) reinterpret_cast<float4*>(dst)[i] = reinterpret_cast<const float4*>(src)[i];
    };
    copy(gimg, img_lds, IMG);
    copy(gwt, lds + LDS_WT, WTB);
    for (int i = threadIdx.x; i < 4 * FIMG_BYTES / 16; i += 256) reinterpret_cast<float4*>(fimg_all)[i] = float4{0.f, 0.f, 0.f, 0.f};
    if (threadIdx.x < 12) reinterpret_cast<uint32_t*>(lds + LDS_READY)[threadIdx.x] = 0u;  // ready[4], done[4], handed[4]
  }
  Geo g;
  g.n = n;
  g.ntiles = (n + 31) / 32;
  g.nchain = (int64_t)gridDim.x * 4;
  g.rounds = (g.ntiles + g.nchain - 1) / g.nchain;  // uniform trip count: every wave takes part in every round's hand-shakes
  uint64_t rng_offset = na.rng_offset;
  if (na.rng_offset_dev) rng_offset += (uint64_t)na.rng_offset_dev[0];  // device-resident part of the counter (hipGraph replays)
#ifdef TF_STAMPS
  uint64_t* stamp_base = reinterpret_cast<uint64_t*>(out);
  out = nullptr;
#endif
  float sq = 0.0f;
  uint32_t gave_up = 0u;
  if (wid < 4) {
    // =============================== chain waves ===============================
    char* my_img = fimg_all + wid * FIMG_BYTES;
    const char* tab = img_lds + (size_t)n_frags<PREC, VAR>() * FB;
    const uint32_t* rec = reinterpret_cast<const uint32_t*>(lds + LDS_HAND) + (wid * 32 + col) * REC_DW;
    const int* ht = reinterpret_cast<const int*>(lds + LDS_HAND + 128 * REC_DW * 4) + wid * 32 + col;
    FimgStoreLane SL = fimg_store_lane(col);
    __syncthreads();  // P: the images and tables of this wave's siblings, round 0's samples of the dW waves
    // A round's inputs: the samples the dW waves handed over (x_t as bf16 pairs, target, timestep) and the timestep's effective-bias
    // row (nine 16-byte gathers from L2, per lane).  They are fetched a round AHEAD -- behind the round's last image, while the dW
    // waves take the layer-0 products -- so the row's L2 round trip (2.5 k cycles at the top of every round when it was fetched
    // there) lies under the dW waves' work, and the forward starts on registers.
    uint32_t xb[5], xbn[5];
    float tg[3];
    int tt;
    float4 bq[9];
    auto load_top = [&]() {
      // (uint4, a struct of four words, on purpose: this hipcc miscompiles __builtin_bit_cast of the ELEMENTS of an ext-vector
      //  load -- every element came back as element 0; so3x_mlp_bwd.hip's zstash_load_layer met the same bug)
      const uint4 r0 = *reinterpret_cast<const uint4*>(rec), r1 = *reinterpret_cast<const uint4*>(rec + 4);
      tt = *ht;
      xbn[0] = r0.x; xbn[1] = r0.y; xbn[2] = r0.z; xbn[3] = r0.w; xbn[4] = r1.x;
      tg[0] = __uint_as_float(r1.y); tg[1] = __uint_as_float(r1.z); tg[2] = __uint_as_float(r1.w);
      const float* beff = beff_tab + (size_t)tt * 96;
#pragma unroll
      for (int q = 0; q < 9; q++) bq[q] = *reinterpret_cast<const float4*>(beff + 8 * q + 4 * h);  // rows 8 q + 4 h ..: tiles 0, 1 and (q = 8) 2
    };
    load_top();
    for (int64_t rd = 0; rd < g.rounds; rd++) {
#pragma unroll
      for (int i = 0; i < 5; i++) xb[i] = xbn[i];
      asm volatile("" : "+v"(SL.rowbase), "+v"(SL.swz8));  // opaque per round: no hoisting of the ~90 store addresses
      // ... and none of the 101 weight-fragment reads: the images are loop-invariant, and hoisted out of the round loop they
      // are 400 registers' worth of spills (seen: 2.4 KB of scratch per lane)
      int lane_r = lane;
      asm volatile("" : "+v"(lane_r));
      const int64_t tile = rd * g.nchain + (int64_t)blockIdx.x * 4 + wid;
      const int64_t s = tile * 32 + col;
      const bool live = tile < g.ntiles && s < n;
      if (wid == 0) TF_STAMP(0, 0);
      typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
      // ---- forward: layer 0 from the per-timestep effective-bias row (K = 9: the rotation entries), then the hidden layers
      uint32_t hpk[4][17], dpk[4][17];
      f32x16 acc[3];
      {
#pragma unroll
        for (int to = 0; to < 3; to++) {
          f32x16 a;
#pragma unroll
          for (int q = 0; q < 4; q++) {
            if (to == 2 && q > 0) {
#pragma unroll
              for (int r = 0; r < 4; r++) a[4 * q + r] = 0.0f;
            } else {
              const float4 v = bq[4 * to + q];
              a[4 * q] = v.x; a[4 * q + 1] = v.y; a[4 * q + 2] = v.z; a[4 * q + 3] = v.w;
            }
          }
          acc[to] = a;
        }
        const bf16x8 b0 = __builtin_bit_cast(bf16x8, u32x4_t{h ? (xb[4] & 0xFFFFu) : xb[0], h ? 0u : xb[1], h ? 0u : xb[2], h ? 0u : xb[3]});
        const bf16x8* w = reinterpret_cast<const bf16x8*>(img_lds);
#pragma unroll
        for (int to = 0; to < 3; to++) acc[to] = mfma_bf16(w[to * 64 + lane_r], b0, acc[to]);
      }
      if (wid == 0) TF_STAMP(0, 1);
      activate_td(acc, hpk[0], dpk[0], h, tab);
      if (wid == 0) TF_STAMP(0, 2);
#pragma unroll
      for (int l = 1; l < 4; l++) {
        Tile<PREC> cur;
        operand_of(hpk[l - 1], cur);
        hidden_fwd_td(img_lds + (size_t)frag_hidden<PREC, VAR>(l) * FB, cur, acc, hpk[l], dpk[l], h, tab, lane_r);
      }
      if (wid == 0) TF_STAMP(0, 3);
      uint32_t seen_done = lds_early(lds, LDS_DONE + 4 * wid);
      f32x16 last[1];
      {
        Tile<PREC> cur;
        operand_of(hpk[3], cur);
        hidden_layer<PREC, 1>(img_lds + (size_t)frag_last<PREC, VAR>() * FB, cur, last, lane_r);
      }
      // ---- MSE and its gradient (diffusion.py:357): head outputs 0..2 = regs 0..2 of the lower half (head_of_row)
      uint32_t pdz[17];
#pragma unroll
      for (int r = 0; r < 17; r++) pdz[r] = 0u;
      if (h == 0) {
        const float d0 = last[0][0] - tg[0], d1 = last[0][1] - tg[1], d2 = last[0][2] - tg[2];
        if (live) {
          sq += d0 * d0 + d1 * d1 + d2 * d2;
          if (out) { out[s * 3] = last[0][0]; out[s * 3 + 1] = last[0][1]; out[s * 3 + 2] = last[0][2]; }
        }
        const float sc = live ? la.dscale : 0.0f;  // dead columns: dZ_4 = 0, hence every dZ_l = 0 and no contribution to any dW sum
        pdz[0] = pack_bf16x2(d0 * sc, d1 * sc);
        pdz[1] = pack_bf16x2(d2 * sc, 0.0f);
      }
      uint4 hq[6];
      // ---- backward, per layer index k = 4 - l:   wait until the dW waves are out of the image's previous filling -- the image's
      // 18 stores INTERLEAVED with the MFMAs of dH_l = W_l^T dZ_l (both read the packed dZ_l; a wave's LDS queue takes a store
      // per 25 cycles, the matrix pipe an MFMA per 32: one hides the other) -- announce the image -- dZ_{l-1} = dH_l * silu'(Z_{l-1})
      f32x16 dh[3];
      // (the dW waves' count for the image about to be refilled is read a phase ahead -- here in front of the head, below in front
      //  of each layer's multiplies -- and looked at in front of the stores: no LDS round trip in the chain's critical path)
      if (wid == 0) TF_STAMP(0, 4);
#pragma unroll
      for (int l = 4; l >= 0; l--) {
        const uint32_t nfill = img_seq(rd, 4 - l);
        char* img_p = my_img;
        asm volatile("" : "+v"(SL.rowbase), "+v"(SL.swz8));  // opaque per layer: the ~20 store addresses are made where they are used
        if (l == 0) {
          // the next round's samples and bias row, IN FRONT of this round's last image: the row's L2 round trip (1 k cycles at the
          // top of the forward when the gathers were issued behind the image) flies under the stores
          lds_wait_ge(LDS_HANDED + 4 * wid, (uint32_t)rd + 1u, gave_up);  // the partner put them in behind this round's first image
          load_top();
        }
        lds_wait_ge_seen(seen_done, LDS_DONE + 4 * wid, 4u * nfill, gave_up);
        if (wid == 0) TF_STAMP(0, 5 + 4 * (4 - l));
        if (l > 0) {
          const uint32_t (&ph)[17] = hpk[l - 1];  // H_l = silu(Z_{l-1}): the forward's operand bits
          // the stores' addresses: chunk 2 C + h of this lane's row sits at rowbase + ((16 C + 8 h) ^ swizzle), and the swizzle
          // touches bits 3..6 only -- eight bases sb[C & 7], the rest of C is an instruction offset (36 address operations
          // per layer otherwise)
          char* sb[8];
#pragma unroll
          for (int c = 0; c < 8; c++) sb[c] = img_p + SL.rowbase + (((2 * c + h) * 8) ^ SL.swz8);
          auto put = [&](int C, uint32_t lo, uint32_t hi) { *reinterpret_cast<uint2*>(sb[C & 7] + (C >> 3) * 128) = uint2{lo, hi}; };
          auto store = [&](int i) {  // store i of the image's 18: the dZ_l block (C = 0..8), then the H_l block (C = 12..20)
            if (i < 8) put(i, pdz[2 * i], pdz[2 * i + 1]);
            else if (i == 8) put(8, h ? 0u : pdz[16], 0u);
            else if (i < 17) put(12 + i - 9, ph[2 * (i - 9)], ph[2 * (i - 9) + 1]);
            else put(20, ph[16], 0u);
          };
          const int KS = l < 4 ? 5 : 1, NM = 3 * KS;
          typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
          bf16x8 bop[5];
#pragma unroll
          for (int ks = 0; ks < 5; ks++) {
            const u32x4 v = ks < 4 ? u32x4{pdz[4 * ks], pdz[4 * ks + 1], pdz[4 * ks + 2], pdz[4 * ks + 3]} : u32x4{pdz[16], 0u, 0u, 0u};
            bop[ks] = __builtin_bit_cast(bf16x8, v);
          }
          // (LDS-resident W^T: a fragment is read three MFMAs ahead of its use, through a four-slot ring -- one wave's LDS operations
          //  complete in order, so a read issued behind the stores would wait for them, and the MFMA for the read)
          bf16x8 wring[4];
          auto wread = [&](int m) {
            wring[m & 3] = reinterpret_cast<const bf16x8*>(lds + LDS_WT)[(size_t)wt_frag<PREC>(l, m / KS, m % KS) * 64 + lane_r];
          };
#pragma unroll
          for (int m = 0; m < 3 && m < NM; m++) wread(m);
#pragma unroll
          for (int m = 0; m < NM; m++) {
            if (m + 3 < NM) wread(m + 3);
#pragma unroll
            for (int i = m * 18 / NM; i < (m + 1) * 18 / NM; i++) store(i);
            __builtin_amdgcn_sched_barrier(0);
            const int to = m / KS, ks = m % KS;
            dh[to] = mfma_bf16(wring[m & 3], bop[ks], ks == 0 ? zero16<PREC>() : dh[to]);
            __builtin_amdgcn_sched_barrier(0);
          }
        } else {  // H_0 = the network input: [0..8] R, [9] one, [10..65] emb(t), zeros; the two lanes of a column split the row
#pragma unroll
          for (int c8 = 0; c8 < 8; c8++) fimg_store_pk(img_p, SL, 2 * c8 + h, pdz[2 * c8], pdz[2 * c8 + 1]);
          fimg_store_pk(img_p, SL, 16 + h, h ? 0u : pdz[16], 0u);
          const uint32_t hd[24] = {hq[0].x, hq[0].y, hq[0].z, hq[0].w, hq[1].x, hq[1].y, hq[1].z, hq[1].w, hq[2].x, hq[2].y, hq[2].z, hq[2].w,
                                   hq[3].x, hq[3].y, hq[3].z, hq[3].w, hq[4].x, hq[4].y, hq[4].z, hq[4].w, hq[5].x, hq[5].y, hq[5].z, hq[5].w};
#pragma unroll
          for (int c4 = 0; c4 < 12; c4++) {
            const int ch0 = 24 + 12 * (lane_r >> 5) + c4;  // chunk of the 4 consecutive input slots 48 h + 4 c4 .. (h from the opaque lane: no hoisting)
            uint32_t lo = hd[2 * c4], hi = hd[2 * c4 + 1];
            if (c4 < 3) {  // the lower half's first three chunks carry the rotation entries and the constant one
              const uint32_t plo = c4 == 0 ? xb[0] : (c4 == 1 ? xb[2] : xb[4]);
              const uint32_t phi = c4 == 0 ? xb[1] : (c4 == 1 ? xb[3] : hi);
              lo = h ? lo : plo;
              hi = h ? hi : phi;
            }
            fimg_store_pk(img_p, SL, ch0, live ? lo : 0u, live ? hi : 0u);
          }
        }
#ifdef TF_FAULT_TEST
        if (!(blockIdx.x == 3 && wid == 1 && rd == 2 && l == 2))
#endif
        lds_post(LDS_READY + 4 * wid, nfill + 1u);
        if (wid == 0) TF_STAMP(0, 6 + 4 * (4 - l));
        if (l > 0) {
          seen_done = lds_early(lds, LDS_DONE + 4 * wid);
          const uint32_t (&dp)[17] = dpk[l - 1];  // silu'(Z_{l-1}), parked by the forward
#pragma unroll
          for (int r = 0; r < 16; r++) {
            // (as fused multiply-adds with a zero addend: v_fma_mix_f32 takes the f16 half as it is -- one instruction, not convert + multiply)
            const float g0 = __builtin_fmaf(r < 8 ? dh[0][2 * r] : dh[1][2 * r - 16], f16_lo(dp[r]), 0.0f);
            const float g1 = __builtin_fmaf(r < 8 ? dh[0][2 * r + 1] : dh[1][2 * r - 15], f16_hi(dp[r]), 0.0f);
            pdz[r] = pack_bf16x2(g0, g1);
          }
          pdz[16] = pack_bf16x2(h ? 0.0f : dh[2][0] * f16_lo(dp[16]), 0.0f);  // upper half of tile 2 / reg 0 = the constant-one row
#pragma unroll
          for (int r = 0; r < 17; r++) asm volatile("" : "+v"(pdz[r]));  // (the packed words, not their fp32 sources, are what lives on)
          // this lane's 48 input slots of the layer-0 image (bf16 bits as the image wants them): six 16-byte gathers, two layers
          // ahead of their store
          if (l == 3) {
#pragma unroll
            for (int i = 0; i < 6; i++) hq[i] = h0_tab[(size_t)tt * 12 + 6 * h + i];
          }
        }
        if (wid == 0) TF_STAMP(0, 7 + 4 * (4 - l));
        if (wid == 0) TF_STAMP(0, 8 + 4 * (4 - l));
      }
    }
  } else {
    // ================================ dW waves =================================
    // distributions.py:42-43: column 0 == sample 0's eps.  With drawn timesteps that is GLOBAL sample 0's draw (Philox index 0,
    // whatever this shard's index_base); with caller-supplied timesteps this call's t[0], as in the reference.
    const int T = na.T;
    int64_t wrow_t = -1;
    if (na.quirk_col0) {
      if (na.t) { const int64_t v = na.t[0]; wrow_t = v < 0 ? 0 : (v >= T ? T - 1 : v); }
      else wrow_t = (int64_t)(((uint64_t)philox4x32_10(na.seed, (uint64_t)0, rng_offset).w * (uint64_t)T) >> 32);
    }
    switch (wid - 4) {
      case 0:
#ifdef TF_STAMPS
        dw_role_fused<0, EXPLICIT>(lds, g, na, rng_offset, wrow_t, x_t_out, slabs, lane, gave_up, stamp_base); break;
#else
        dw_role_fused<0, EXPLICIT>(lds, g, na, rng_offset, wrow_t, x_t_out, slabs, lane, gave_up); break;
#endif
      case 1: dw_role_fused<1, EXPLICIT>(lds, g, na, rng_offset, wrow_t, x_t_out, slabs, lane, gave_up); break;
      case 2: dw_role_fused<2, EXPLICIT>(lds, g, na, rng_offset, wrow_t, x_t_out, slabs, lane, gave_up); break;
      default: dw_role_fused<3, EXPLICIT>(lds, g, na, rng_offset, wrow_t, x_t_out, slabs, lane, gave_up); break;
    }
  }
  // ---- the reported loss: per-block sums of squares combined by the last block to arrive, in a fixed tree (deterministic)
  double* wsum = reinterpret_cast<double*>(lds + LDS_RED);
  int* is_last = reinterpret_cast<int*>(lds + LDS_RED + 64);
  double v = (double)sq;
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
  if (gave_up) v = __builtin_nan("");  // a hand-shake of this wave timed out: the step's loss says so
  if (lane == 0) wsum[wid] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    // (the dW waves' entries are zeros -- or the NaN of a timed-out hand-shake: x + 0.0 * 0.0 is x, bit for bit)
    const double bs = ((wsum[0] + wsum[1]) + (wsum[2] + wsum[3])) + 0.0 * ((wsum[4] + wsum[5]) + (wsum[6] + wsum[7]));
    __hip_atomic_store(la.partial + blockIdx.x, bs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *is_last = last_block_arrives(la.ticket) ? 1 : 0;
  }
  __syncthreads();  // the other waves of the last block read the partials only behind this barrier
  if (*is_last) {
    double a = 0.0;
    for (unsigned b = threadIdx.x; b < gridDim.x; b += 512) a += __hip_atomic_load(la.partial + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) a += __shfl_xor(a, m);
    __syncthreads();
    if (lane == 0) wsum[wid] = a;
    __syncthreads();
    if (threadIdx.x == 0) {
      la.loss[0] = (float)((((wsum[0] + wsum[1]) + (wsum[2] + wsum[3])) + ((wsum[4] + wsum[5]) + (wsum[6] + wsum[7]))) * la.inv_count);
      if (la.status) la.status[0] = la.loss[0];
      if (la.rng_counter) la.rng_counter[0] += 1;  // every block read this step's offset at its start
    }
  }
#ifdef TF_STAMPS
  if (blockIdx.x == 0 && (wid == 0 || wid == 4) && lane == 0) {   // [role][round 63]: kernel entry / exit in shader ticks and in 100 MHz ticks
    uint64_t* e = stamp_base + ((wid >> 2) * 64 + 63) * 32;
    e[0] = t_entry; e[1] = __builtin_amdgcn_s_memtime(); e[2] = rt_entry; e[3] = __builtin_amdgcn_s_memrealtime();
  }
#endif
}

}  // namespace

extern "C" {

int so3x_train_fused(so3x_stream_t s, const float* params, const float* sched, int T, const float* trap_q, const uint16_t* guide_q,
                     const float* x0, const int64_t* t, int64_t* t_used, int quirk_col0, const float* axes, const float* unif, uint64_t seed,
                     uint64_t rng_offset, int64_t* rng_counter, int64_t index_base, int64_t n, float* loss, float* x_t, float* out,
                     void* workspace, size_t workspace_bytes) {
  if (n <= 0 || T <= 0 || !params || !sched || !trap_q || !x0 || !loss || ((axes == nullptr) != (unif == nullptr))) return SO3X_ERR_INVALID_ARG;
  const TrainLayout L = train_layout(n, T);
  if (!workspace || workspace_bytes < L.end) return SO3X_ERR_WORKSPACE;
  char* ws = (char*)workspace;
  hipStream_t st = (hipStream_t)s;
  // one prep launch: forward image (with the (silu, silu') table), transposed image, per-timestep tables; clears the ticket
  int rc = launch_prep(st, params, PREC, VAR, T, ws, 3, (void*)(ws + L.wt), true, reinterpret_cast<unsigned*>(ws + L.ticket));
  if (rc) return rc;
  int64_t* counter = (axes == nullptr || t == nullptr) ? rng_counter : nullptr;  // advanced only by a call that drew from it
  LossArgs la;
  la.target = nullptr; la.dout = nullptr; la.loss = loss;
  la.partial = reinterpret_cast<double*>(ws + L.partial);
  la.status = reinterpret_cast<float*>(ws + L.ticket + 64);
  la.ticket = reinterpret_cast<unsigned*>(ws + L.ticket);
  la.rng_counter = counter;
  la.dscale = (float)(2.0 / (3.0 * (double)n));
  la.inv_count = 1.0 / (3.0 * (double)n);
  NoiseArgs na{sched, trap_q, guide_q, x0, t, t_used, axes, unif, rng_counter, seed, rng_offset, index_base, T, quirk_col0};
  const int64_t nt = (n + 31) / 32;
  const int gf = (int)((nt + 3) / 4 < DW_BLOCKS ? (nt + 3) / 4 : DW_BLOCKS);  // = the slab count so3x_train_bwd_reduce sums
  const float* beff = reinterpret_cast<const float*>(ws + beff_offset(PREC, GATHER));
  const uint4* h0 = reinterpret_cast<const uint4*>(ws + h0_offset(PREC, GATHER, T));
  float* slabs = reinterpret_cast<float*>(ws + L.slabs);
  static PerDevice attr_e, attr_p;
  if (axes) {
    if ((rc = ensure_dyn_lds(attr_e, reinterpret_cast<const void*>(&k_train_fused<true>), LDS_TOTAL))) return rc;
    hipLaunchKernelGGL((k_train_fused<true>), dim3(gf), dim3(512), LDS_TOTAL, st, (const void*)ws, (const void*)(ws + L.wt), beff, h0, na, x_t, out,
                       slabs, n, la);
  } else {
    if ((rc = ensure_dyn_lds(attr_p, reinterpret_cast<const void*>(&k_train_fused<false>), LDS_TOTAL))) return rc;
    hipLaunchKernelGGL((k_train_fused<false>), dim3(gf), dim3(512), LDS_TOTAL, st, (const void*)ws, (const void*)(ws + L.wt), beff, h0, na, x_t, out,
                       slabs, n, la);
  }
  return check_launch();
}

}  // extern "C"
