// so3x_mlp.hpp -- the RotPredict score network (so3_train.py:11-49, models.py:13-25)
// on the CDNA4 matrix cores, shared by so3x_mlp.hip (standalone fwd/bwd) and
// so3x_diffusion.hip (fused reverse chain).
//
// Orientation: H_{l+1}[feature, sample] = W_l . H_l with the
// WEIGHTS as the MFMA A operand and the ACTIVATIONS as the B operand.  A 32x32 result
// tile then has its sample on the lane (col = lane & 31) and its 32 features in the 16
// accumulator registers of the two lane halves:
//       row(reg, h) = (reg & 3) + 8 * (reg >> 2) + 4 * h,     h = lane >> 5.
// The next layer contracts over the feature (row) index, so an accumulator tile is
// consumed as the next B operand with NO lane movement and NO LDS round trip:
//   * fp32  (v_mfma_f32_32x32x2_f32):  k-step = one accumulator register; lane half h
//           supplies k = h, i.e. feature row(reg, h).
//   * bf16  (v_mfma_f32_32x32x16_bf16): k-step s of a tile = registers 8s..8s+7 packed
//           to bf16; element j of half h is feature 16s + 8(j>>2) + 4h + (j&3).
// The weight fragments are laid out once per launch, by a prep kernel, in exactly that
// permuted k order (the "image"), copied to LDS by every workgroup and read with one
// ds_read per MFMA.
//
// Feature space of a hidden activation: 96 rows = 3 tiles.  Rows 0..64 are the 65
// features, row 68 (tile 2, reg 0, upper lane half) is forced to 1.0 after the
// activation and carries the next layer's bias as an ordinary weight column; all other
// rows of tile 2 are zero.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/so3x.h"
#include "so3x_math.hpp"

namespace so3x {
namespace mlp {

constexpr int D = 65;
constexpr int NEMB = 56;   // 28 sines then 28 cosines, models.py:24
constexpr int NFREQ = 28;
constexpr int ONE_ROW = 68;  // hidden-feature row that carries the constant 1
constexpr int LAYER_STRIDE = D * D + D;
constexpr int NPARAMS = 4 * LAYER_STRIDE + 3 * D + 3;
static_assert(NPARAMS == SO3X_MLP_PARAMS, "param count");
// The output layer is 3 wide (out_type "skewvec") or 6 wide ("rotmat", so3_train.py:19-22); n_out is a run-time argument.
// Head output o sits in row o (o < 4) or o + 4 of the layer's single 32-row tile: rows 0..3, 8, 9 are accumulator
// registers 0..5 of the LOWER lane half, so one lane holds all of a sample's outputs, and -- read as a contraction index
// (dH_4 = W_4^T dout) -- K slots 0..5 of that half.
constexpr int NOUT_MAX = 6;
constexpr int NPARAMS_MAX = 4 * LAYER_STRIDE + NOUT_MAX * D + NOUT_MAX;  // slab stride of the dW partials
static_assert(NPARAMS_MAX == SO3X_MLP_PARAMS_ROTMAT, "param count");
__host__ __device__ constexpr int nparams(int nout) { return 4 * LAYER_STRIDE + nout * (D + 1); }
__host__ __device__ constexpr int head_of_row(int row) { return row < 4 ? row : ((row == 8 || row == 9) ? row - 4 : -1); }

struct Freqs { float f[NFREQ]; };  // passed by value as a kernel argument

using f32x16 = float __attribute__((ext_vector_type(16)));
using bf16x8 = __bf16 __attribute__((ext_vector_type(8)));

__host__ __device__ constexpr int row_of(int reg, int h) { return (reg & 3) + 8 * (reg >> 2) + 4 * h; }

// ---- image geometry -------------------------------------------------------------
// variant CHAIN: layer 0 sees only the 9 rotation entries (the 56 time-embedding
// inputs are folded into a per-timestep effective bias, appendix C.3);
// variant FULL : layer 0 sees [R(9), 1, emb(56)] per sample (per-sample timesteps).
enum Variant { CHAIN = 0, FULL = 1, GATHER = 2, GATHER_T = 3, GATHER_TD = 4 };
// GATHER: the CHAIN layer-0 layout (K = 9, per-timestep effective bias) with a PER-SAMPLE bias row
// gathered by t from the [T][96] table -- per-sample timesteps without evaluating 56 sin/cos per sample.
// Used by the backward when the caller bounds t (t_table > 0); never scale-folded (true pre-activations).
// GATHER_T: GATHER with the SiLU table's scale fold (bf16): the training forward.  Its MFMAs emit u = 16 z + 127.5; the
// pre-activation the backward wants goes to the stash as z = (u - 127.5) / 16 (one multiply-add per value).
// GATHER_TD: GATHER_T with 16-byte table entries (alpha, beta, alpha', beta'): one lookup yields silu(z) AND silu'(z) -- the
// forward of the one-kernel training step (so3x_train_fused.hip) parks the derivative instead of the pre-activation.
__host__ __device__ constexpr bool chain_layout(int var) { return var != FULL; }
__host__ __device__ constexpr bool gather_layout(int var) { return var == GATHER || var == GATHER_T || var == GATHER_TD; }

// number of k-steps
template <int PREC> __host__ __device__ constexpr int ks_hidden() { return PREC == SO3X_PREC_F32 ? 33 : 5; }
template <int PREC, int VAR> __host__ __device__ constexpr int ks_layer0() {
  return PREC == SO3X_PREC_F32 ? (chain_layout(VAR) ? 5 : 33) : (chain_layout(VAR) ? 1 : 5);
}
// a fragment = what the 64 lanes read for one MFMA: 64 x 4 B (fp32) or 64 x 16 B (bf16)
template <int PREC> __host__ __device__ constexpr int frag_bytes() { return PREC == SO3X_PREC_F32 ? 256 : 1024; }
template <int PREC, int VAR> __host__ __device__ constexpr int frag_l0() { return 0; }
template <int PREC, int VAR> __host__ __device__ constexpr int frag_hidden(int l /*1..3*/) {
  return 3 * ks_layer0<PREC, VAR>() + (l - 1) * 3 * ks_hidden<PREC>();
}
template <int PREC, int VAR> __host__ __device__ constexpr int frag_last() { return frag_hidden<PREC, VAR>(4); }
template <int PREC, int VAR> __host__ __device__ constexpr int n_frags() { return frag_last<PREC, VAR>() + ks_hidden<PREC>(); }
// (the SiLU table of the folded bf16 CHAIN variant rides behind the fragments: see fold_scale below)
template <int PREC, int VAR> __host__ __device__ constexpr int image_bytes() {
  return n_frags<PREC, VAR>() * frag_bytes<PREC>() +
         (PREC != SO3X_PREC_BF16 ? 0 : ((VAR == 0 /*CHAIN*/ || VAR == 3 /*GATHER_T*/) ? 2048 : (VAR == 4 /*GATHER_TD*/ ? 4096 : 0)));
}

// hidden feature index fed by (k-step ks, lane half h, element j)
template <int PREC> __host__ __device__ inline int hidden_feature(int ks, int h, int j) {
  if (PREC == SO3X_PREC_F32) {
    const int tin = ks < 16 ? 0 : (ks < 32 ? 1 : 2);
    const int reg = ks < 16 ? ks : (ks < 32 ? ks - 16 : 0);
    return 32 * tin + row_of(reg, h);
  } else {
    const int tin = ks >> 1, s = ks & 1;  // ks 0..4 -> (0,0)(0,1)(1,0)(1,1)(2,0)
    return 32 * tin + 16 * s + 8 * (j >> 2) + 4 * h + (j & 3);
  }
}
// layer-0 input slot fed by (ks, h, j): fp32 slot = 2 ks + h, bf16 slot = 16 ks + 8 h + j
template <int PREC> __host__ __device__ inline int l0_slot(int ks, int h, int j) {
  return PREC == SO3X_PREC_F32 ? 2 * ks + h : 16 * ks + 8 * h + j;
}
// slot -> column of net.0.weight (0..64), -2 = bias (constant-one input), -1 = zero padding.
//  fp32 FULL: [0..8] R, [9] one, [10..65] emb.   bf16 FULL: [0..8] R, [9] one, [16..71] emb.
template <int PREC, int VAR> __host__ __device__ inline int l0_slot_to_col(int slot) {
  if (slot < 9) return slot;
  if (chain_layout(VAR)) return -1;
  if (slot == 9) return -2;
  const int e0 = PREC == SO3X_PREC_F32 ? 10 : 16;
  if (slot >= e0 && slot < e0 + NEMB) return 9 + (slot - e0);
  return -1;
}
template <int PREC> __host__ __device__ constexpr int l0_emb_slot0() { return PREC == SO3X_PREC_F32 ? 10 : 16; }

// SiLU from an LDS table (bf16 sampling / inference path: CHAIN variant).  A SiLU as exp2, add, rcp, mul costs 24 cycles of
// the SIMD's vector issue port per value and there are 264 values per lane per reverse step: two thirds of the chain
// kernel.  Instead the weight image makes every hidden layer's MFMAs emit the table coordinate directly,
//       u = 16 z + 127.5        (weights x 16; the bias column carries 16 b; a second constant-one row, 69, carries 127.5),
// and the activation is   h = alpha_i + beta_i u,   i = sat_u8(rne(u))   -- one v_cvt_pk_u8_f32 (saturating on both sides), one
// shift-add for the byte address, one ds_read_b64 of (alpha_i, beta_i), one v_fma: THREE vector instructions and no
// transcendental.  Entry i holds the line through silu over z in [(i - 128)/16, (i - 127)/16), shifted to halve its one-sided
// error (max |error| 1.3e-4, a tenth of a bf16 ulp at 0.25; the end entries continue silu's asymptotes 0 and z).  h is the
// TRUE activation, so the next layer's weights are the plain ones (x 16 again for its own table coordinate).  The 2 KB table
// rides at the end of the weight image and reaches LDS with it.  Measured on the chain kernel (profiles/r02_ab_chain_*.json):
// -11 % time with a 4-instruction form of this, against +3 % LDS-port pressure feared; the LDS port has the room.
// The chain kernel goes one further (wide_tab_entry below): a lane-replicated copy of the table at LDS address 0 turns the
// convert's result into the address itself -- TWO vector instructions, and lookups that cannot meet in a bank.
constexpr float kTabC = 16.0f, kTabD = 127.5f;
constexpr int kSiluTabEntries = 256, kSiluTabBytes = kSiluTabEntries * 8;
constexpr int ONE_ROW2 = 69;  // second constant-one hidden row (tile 2, reg 1 of the upper lane half): carries kTabD
static_assert(CHAIN == 0 && GATHER_T == 3 && GATHER_TD == 4 && kSiluTabBytes == 2048, "image_bytes() above spells these out");
template <int PREC, int VAR> __host__ __device__ constexpr bool fold_scale() {
  return PREC == SO3X_PREC_BF16 && (VAR == CHAIN || VAR == GATHER_T || VAR == GATHER_TD);
}

// weight-image element value: fragment `frag`, lane, element j (bf16 only)
template <int PREC, int VAR>
__device__ inline float image_value(const float* __restrict__ params, int frag, int lane, int j, int nout) {
  const int i = lane & 31, h = lane >> 5;
  int l, tout, ks;
  constexpr int K0 = ks_layer0<PREC, VAR>(), KH = ks_hidden<PREC>();
  if (frag < 3 * K0) { l = 0; tout = frag / K0; ks = frag % K0; }
  else if (frag < frag_last<PREC, VAR>()) { int f = frag - 3 * K0; l = 1 + f / (3 * KH); f %= 3 * KH; tout = f / KH; ks = f % KH; }
  else { l = 4; tout = 0; ks = frag - frag_last<PREC, VAR>(); }
  const int o = l < 4 ? 32 * tout + i : head_of_row(i);
  if (o < 0 || o >= (l < 4 ? D : nout)) return 0.0f;
  const float* W = params + l * LAYER_STRIDE;
  const float* bias = W + (l < 4 ? D : nout) * D;
  if (l == 0) {
    const int col = l0_slot_to_col<PREC, VAR>(l0_slot<PREC>(ks, h, j));
    const float sc0 = fold_scale<PREC, VAR>() ? kTabC : 1.0f;
    return sc0 * (col >= 0 ? W[o * D + col] : (col == -2 ? bias[o] : 0.0f));
  }
  const int f = hidden_feature<PREC>(ks, h, j);
  const float sc = (fold_scale<PREC, VAR>() && l < 4) ? kTabC : 1.0f;  // hidden layers emit u = 16 z + 127.5; the head emits its outputs
  if (f == ONE_ROW2) return (fold_scale<PREC, VAR>() && l < 4) ? kTabD : 0.0f;
  return f < D ? sc * W[o * D + f] : (f == ONE_ROW ? sc * bias[o] : 0.0f);
}

// entry i of the SiLU table (alpha, beta): h = alpha + beta u on u in [i - 0.5, i + 0.5)
__device__ inline float2 silu_table_entry(int i) {
  const float h = 1.0f / kTabC, z0 = ((float)i - 128.0f) * h, z1 = z0 + h, zm = z0 + 0.5f * h;
  auto f = [](float z) { return z / (1.0f + expf(-z)); };
  float bz, az;
  if (i == 0) { bz = 0.0f; az = f(zm); }                                   // z < -7.94: silu -> 0
  else if (i == kSiluTabEntries - 1) { bz = 1.0f; az = f(z0) - z0; }        // z >= 7.94: silu -> z
  else {
    bz = (f(z1) - f(z0)) * kTabC;
    az = f(z0) - bz * z0;
    az += 0.5f * (f(zm) - (az + bz * zm));                                  // centre the secant's one-sided error
  }
  return float2{az - bz * kTabD / kTabC, bz / kTabC};                        // in terms of u = 16 z + 127.5
}

// entry i of the GATHER_TD table: (alpha, beta) as above and (alpha', beta') with silu'(z) = alpha' + beta' u on the same
// interval -- the secant of silu' = sigma (1 + z (1 - sigma)), centred like silu's (max |error| 1.2e-4; silu' is parked as f16)
__device__ inline float4 silu_table_entry4(int i) {
  const float2 e = silu_table_entry(i);
  const float h = 1.0f / kTabC, z0 = ((float)i - 128.0f) * h, z1 = z0 + h, zm = z0 + 0.5f * h;
  auto g = [](float z) { const float sg = 1.0f / (1.0f + expf(-z)); return sg * (1.0f + z * (1.0f - sg)); };
  float bz, az;
  if (i == 0) { bz = 0.0f; az = g(zm); }                                   // z < -7.94: silu' -> 0
  else if (i == kSiluTabEntries - 1) { bz = 0.0f; az = g(zm); }             // z >= 7.94: silu' -> 1
  else {
    bz = (g(z1) - g(z0)) * kTabC;
    az = g(z0) - bz * z0;
    az += 0.5f * (g(zm) - (az + bz * zm));
  }
  return float4{e.x, e.y, az - bz * kTabD / kTabC, bz / kTabC};
}

// ---- transposed-weight image (A operand of dH = W^T dZ in the backward), global/L2-resident -----------
// fragment order: layers 1..3: [l-1][To(in-feature tile) 3][ks over out-features KH], then layer 4: [To 3][K4]
template <int PREC> __host__ __device__ constexpr int k4() { return PREC == SO3X_PREC_F32 ? NOUT_MAX : 1; }  // k-steps covering head slots 0..5
template <int PREC> __host__ __device__ constexpr int wt_frag(int l, int to, int ks) {
  return l < 4 ? ((l - 1) * 3 + to) * ks_hidden<PREC>() + ks : 9 * ks_hidden<PREC>() + to * k4<PREC>() + ks;
}
template <int PREC> __host__ __device__ constexpr int wt_nfrags() { return 9 * ks_hidden<PREC>() + 3 * k4<PREC>(); }
template <int PREC> __device__ inline float wt_value(const float* __restrict__ params, int frag, int lane, int j, int nout) {
  constexpr int KH = ks_hidden<PREC>();
  const int i = lane & 31, h = lane >> 5;
  int l, to, ks;
  if (frag < 9 * KH) { l = 1 + frag / (3 * KH); to = (frag % (3 * KH)) / KH; ks = frag % KH; }
  else { const int f = frag - 9 * KH; l = 4; to = f / k4<PREC>(); ks = f % k4<PREC>(); }
  const int in = 32 * to + i;                       // row of W^T = input feature of layer l
  int out = hidden_feature<PREC>(ks, h, j);         // k index = output feature of layer l (layer 4: its tile row)
  if (l == 4) out = head_of_row(out);
  return (in < D && out >= 0 && out < (l < 4 ? D : nout)) ? params[l * LAYER_STRIDE + out * D + in] : 0.0f;
}

// ---- activations ------------------------------------------------------------------
template <int PREC> __device__ __forceinline__ float silu(float x) {
  if (PREC == SO3X_PREC_F32) return x * sigmoid_f32(x);                     // fp32 path (parity gate G5)
  return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x));                      // v_exp_f32 + v_rcp_f32
}

// time embedding element e of timestep t: models.py:22-24 -- the angle is formed in fp32
// (int64 t promoted to fp32, times the fp32 frequency), then sin / cos of that fp32 angle.
__device__ __forceinline__ float emb_value(int64_t t, int e, const Freqs& fr) {
  const float a = (float)t * fr.f[e < NFREQ ? e : e - NFREQ];
  float sn, cs;
  sincos_cw(a, &sn, &cs);
  return e < NFREQ ? sn : cs;
}

// Per-wave state of one 32-sample tile flowing through the network.
template <int PREC> struct Tile;

template <> struct Tile<SO3X_PREC_F32> {
  float h[3][16];  // hidden activations in accumulator layout (tile 2: only reg 0 is live)
};
template <> struct Tile<SO3X_PREC_BF16> {
  bf16x8 b[5];  // the 5 packed k-steps of the next layer's B operand
};

template <int PREC> __device__ __forceinline__ f32x16 zero16() {
  return f32x16{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};  // folds into the MFMA's inline-0 C operand
}

__device__ __forceinline__ f32x16 mfma_f32(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mfma_bf16(bf16x8 a, bf16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
// F16 = the f16-operand leg of the paired chain stream (SO3X_PREC_F16; so3x_p_sample_chain only): the SAME image layout, 16-bit
// slots and registers, but the slots hold IEEE half bits and the products run on v_mfma_f32_32x32x16_f16 (same rate as bf16).
// What it buys: an activation's multiply-add, conversion and pack become ONE instruction each half (v_fma_mixlo / mixhi_f16)
// instead of fma + half a v_cvt_pk_bf16_f32; f16 carries three more mantissa bits than bf16.  bf16x8 stays the container type.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
template <bool F16> __device__ __forceinline__ f32x16 mfma_op(bf16x8 a, bf16x8 b, f32x16 c) {
  if constexpr (F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
template <bool F16> __device__ __forceinline__ uint32_t pack_pair(float a, float b) {
  if constexpr (F16) {
    typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(uint32_t, h2_t{(_Float16)a, (_Float16)b});
  } else {
    typedef __bf16 b2_t __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(uint32_t, b2_t{(__bf16)a, (__bf16)b});
  }
}
template <bool F16> __device__ __forceinline__ bf16x8 pack_octet(const float* v) {
  typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
  return __builtin_bit_cast(bf16x8, u32x4_t{pack_pair<F16>(v[0], v[1]), pack_pair<F16>(v[2], v[3]), pack_pair<F16>(v[4], v[5]), pack_pair<F16>(v[6], v[7])});
}

// activation + repack of the three accumulator tiles into the next layer's operand
template <int PREC, bool FOLD = false> __device__ __forceinline__ void activate(const f32x16 (&acc)[3], Tile<PREC>& out, int h,
                                                                                const char* tab = nullptr);

// u = 16 z + 127.5  ->  silu(z) from the (alpha, beta) table at `tab` (LDS)
__device__ __forceinline__ float silu_tab(float u, const char* tab) {
  const unsigned idx = __builtin_amdgcn_cvt_pk_u8_f32(u, 0u, 0u);  // round to nearest, saturated to 0..255
  const float2 e = *reinterpret_cast<const float2*>(tab + idx * 8);
  return fmaf(e.y, u, e.x);
}

// Lane-replicated form of the table (the chain kernel): entry i sits 32 times in the 256-byte row i of a 64 KB block that
// starts at LDS address 0, copy c at byte 8 c.  v_cvt_pk_u8_f32 can drop its saturated byte into ANY byte of a third operand:
// with byte 1 of the per-lane constant 8 (lane & 31) it yields the entry's ADDRESS in the one instruction -- no shift-add --
// and the lookups of a wave can never meet in a bank (lane l always reads banks 2 (l & 31), 2 (l & 31) + 1, whatever its
// index).  TWO vector instructions per activation (convert, multiply-add) plus the packing.
constexpr int kWideTabBytes = kSiluTabEntries * 256;
__device__ __forceinline__ uint32_t wide_tab_lane(int lane) { return 8u * (lane & 31); }
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) f32x2_t* lds_f2_ptr;
__device__ __forceinline__ float2 wide_tab_entry(float u, uint32_t lt) {
  const uint32_t addr = __builtin_amdgcn_cvt_pk_u8_f32(u, 1u, lt);  // round to nearest, saturated to 0..255, into bits 8..15
  const f32x2_t e = *(lds_f2_ptr)(uintptr_t)addr;
  return float2{e[0], e[1]};
}
__device__ __forceinline__ float silu_tabw(float u, uint32_t lt) {
  const float2 e = wide_tab_entry(u, lt);
  return fmaf(e.y, u, e.x);
}
// fills the block at LDS address 0 from the 2 KB table of a weight image (global); the caller synchronises
__device__ __forceinline__ void fill_wide_tab(const void* __restrict__ narrow) {
  const float2* src = reinterpret_cast<const float2*>(narrow);
  for (int e = threadIdx.x; e < kSiluTabEntries * 32; e += blockDim.x) {
    const float2 v = src[e >> 5];
    *(lds_f2_ptr)(uintptr_t)((e >> 5) * 256 + (e & 31) * 8) = f32x2_t{v.x, v.y};
  }
}

template <> __device__ __forceinline__ void activate<SO3X_PREC_F32, false>(const f32x16 (&acc)[3], Tile<SO3X_PREC_F32>& out, int h, const char*) {
#pragma unroll
  for (int t = 0; t < 2; t++)
#pragma unroll
    for (int r = 0; r < 16; r++) out.h[t][r] = silu<SO3X_PREC_F32>(acc[t][r]);
  out.h[2][0] = h ? 1.0f : silu<SO3X_PREC_F32>(acc[2][0]);  // row 64 (h=0) / the constant-one row 68 (h=1)
}

// Table form (FOLD): the lookups of 16 values are issued together and consumed afterwards, so that the LDS latency
// (~100 cycles) is paid once per group under counted waits instead of once per four values (what the compiler's own
// schedule of the value-by-value loop does).
template <bool FOLD, bool WIDE = false, bool F16 = false>
__device__ __forceinline__ void activate_bf16(const f32x16 (&acc)[3], Tile<SO3X_PREC_BF16>& out, int h, const char* tab, uint32_t lt = 0) {
  if constexpr (FOLD) {
    constexpr int G = 16;  // 8, 16 or 32 values per group (32 = both k-steps of a tile pair)
    float val[32];
#pragma unroll
    for (int g0 = 0; g0 < 32; g0 += G) {
      float2 e[G];
      float u[G];
#pragma unroll
      for (int i = 0; i < G; i++) {
        const int q = g0 + i;
        u[i] = acc[q >> 4][q & 15];
        if constexpr (WIDE) {
          e[i] = wide_tab_entry(u[i], lt);
        } else {
          const unsigned idx = __builtin_amdgcn_cvt_pk_u8_f32(u[i], 0u, 0u);
          e[i] = *reinterpret_cast<const float2*>(tab + idx * 8);
        }
      }
#pragma unroll
      for (int i = 0; i < G; i++) val[g0 + i] = fmaf(e[i].y, u[i], e[i].x);
    }
#pragma unroll
    for (int t = 0; t < 2; t++)
#pragma unroll
      for (int s = 0; s < 2; s++) out.b[2 * t + s] = pack_octet<F16>(&val[16 * t + 8 * s]);
    const float last8[8] = {h ? 1.0f : (WIDE ? silu_tabw(acc[2][0], lt) : silu_tab(acc[2][0], tab)),   // row 64 | the constant-one row 68
                            h ? 1.0f : 0.0f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};                              // row 69: the second constant one (table offset)
    out.b[4] = pack_octet<F16>(last8);
  } else {
#pragma unroll
    for (int t = 0; t < 2; t++)
#pragma unroll
      for (int s = 0; s < 2; s++) {
        bf16x8 p;
#pragma unroll
        for (int j = 0; j < 8; j++) p[j] = (__bf16)silu<SO3X_PREC_BF16>(acc[t][8 * s + j]);
        out.b[2 * t + s] = p;
      }
    bf16x8 p;
#pragma unroll
    for (int j = 0; j < 8; j++) p[j] = (__bf16)0.0f;
    p[0] = (__bf16)(h ? 1.0f : silu<SO3X_PREC_BF16>(acc[2][0]));  // row 64 | the constant-one row 68
    out.b[4] = p;
  }
}
template <> __device__ __forceinline__ void activate<SO3X_PREC_BF16, false>(const f32x16 (&acc)[3], Tile<SO3X_PREC_BF16>& out, int h, const char* tab) {
  activate_bf16<false>(acc, out, h, tab);
}
template <> __device__ __forceinline__ void activate<SO3X_PREC_BF16, true>(const f32x16 (&acc)[3], Tile<SO3X_PREC_BF16>& out, int h, const char* tab) {
  activate_bf16<true>(acc, out, h, tab);
}

// One hidden layer (NT output tiles) from LDS-resident weight fragments.
template <int PREC, int NT>
__device__ __forceinline__ void hidden_layer(const char* __restrict__ wl /*LDS, first fragment of the layer*/,
                                             const Tile<PREC>& in, f32x16 (&acc)[NT], int lane) {
  __builtin_amdgcn_sched_barrier(0);  // keep this layer's LDS weight reads from being hoisted above the previous layer (else ~240 VGPR spills)
  if constexpr (PREC == SO3X_PREC_F32) {
    const float* w = reinterpret_cast<const float*>(wl);
#pragma unroll
    for (int to = 0; to < NT; to++) {
      f32x16 a = zero16<PREC>();
#pragma unroll
      for (int ks = 0; ks < 33; ks++) {
        const float b = ks < 16 ? in.h[0][ks] : (ks < 32 ? in.h[1][ks - 16] : in.h[2][0]);
        a = mfma_f32(w[(to * 33 + ks) * 64 + lane], b, a);
      }
      acc[to] = a;
    }
  } else {
    const bf16x8* w = reinterpret_cast<const bf16x8*>(wl);
#pragma unroll
    for (int to = 0; to < NT; to++) {
      if (to == 2) __builtin_amdgcn_sched_barrier(0);  // cap the weight-fragment prefetch at two output tiles (40 VGPRs)
      f32x16 a = zero16<PREC>();
#pragma unroll
      for (int ks = 0; ks < 5; ks++) a = mfma_bf16(w[(to * 5 + ks) * 64 + lane], in.b[ks], a);
      acc[to] = a;
    }
  }
}

// Layer 0, CHAIN variant: accumulators start from the per-timestep effective bias
// beff[96] (global/L2, identical address across a lane half -> broadcast load) and add
// W_0[:, 0:9] . R.   XSRC says where the 9 rotation entries of this lane's sample column are:
//   0: x[] already holds the column's sample in both lanes of the column (standalone forward);
//   1: wave64 tile A -- x[] is the lane's OWN rotation, column c lives in lane c      (lower half owns);
//   2: wave64 tile B -- x[] is the lane's OWN rotation, column c lives in lane 32 + c (upper half owns).
// For 1/2 only the entries the other half actually feeds to the MFMA are exchanged (ds_bpermute):
// bf16 1 + 8 values per step, fp32 4 + 5 -- instead of mirroring all 9 twice.
template <int PREC, int XSRC = 0>
__device__ __forceinline__ void layer0_chain(const char* __restrict__ wl, const float* __restrict__ beff, const float* x,
                                             f32x16 (&acc)[3], int lane) {
  const int h = lane >> 5;
  // value of entry j as seen by the lane half that feeds it: `feeder_is_upper` = the MFMA slot belongs to h == 1
  auto entry = [&](int j, bool feeder_is_upper) -> float {
    if (XSRC == 0) return x[j];
    const bool owner_is_upper = (XSRC == 2);
    return feeder_is_upper == owner_is_upper ? x[j] : __shfl_xor(x[j], 32);
  };
#pragma unroll
  for (int to = 0; to < 3; to++) {
    f32x16 a;
#pragma unroll
    for (int q = 0; q < 4; q++) {
      if (to == 2 && q > 0) {
#pragma unroll
        for (int r = 0; r < 4; r++) a[4 * q + r] = 0.0f;
      } else {
        const float4 v = *reinterpret_cast<const float4*>(beff + 32 * to + 8 * q + 4 * h);
        a[4 * q] = v.x; a[4 * q + 1] = v.y; a[4 * q + 2] = v.z; a[4 * q + 3] = v.w;
      }
    }
    acc[to] = a;
  }
  if constexpr (PREC == SO3X_PREC_F32) {
    const float* w = reinterpret_cast<const float*>(wl);
    float xs[5];
#pragma unroll
    for (int m = 0; m < 5; m++) {  // slot 2m + h; slot 9 is padding
      const float lo = entry(2 * m, false), hi = (2 * m + 1 < 9) ? entry((2 * m + 1) % 9, true) : 0.0f;
      xs[m] = h ? hi : lo;
    }
#pragma unroll
    for (int to = 0; to < 3; to++)
#pragma unroll
      for (int m = 0; m < 5; m++) acc[to] = mfma_f32(w[(to * 5 + m) * 64 + lane], xs[m], acc[to]);
  } else {
    const bf16x8* w = reinterpret_cast<const bf16x8*>(wl);
    bf16x8 b;
    // the exchanges are cross-lane operations: evaluate them in uniform control flow, select afterwards
    const float x8 = entry(8, true);
    float xe[8];
#pragma unroll
    for (int j = 0; j < 8; j++) xe[j] = entry(j, false);
#pragma unroll
    for (int j = 0; j < 8; j++) b[j] = (__bf16)(h ? (j == 0 ? x8 : 0.0f) : xe[j]);  // slot 8h + j
#pragma unroll
    for (int to = 0; to < 3; to++) acc[to] = mfma_bf16(w[to * 64 + lane], b, acc[to]);
  }
}

// Layer 0 of the bf16 chain kernel with PER-TIMESTEP A fragments: the effective bias of timestep t rides in three extra K
// slots of the layer's single k-step (slot 9 = bf16(16 beff), slot 10 = bf16 of the remainder: 16 significant bits; slot 11 =
// the table offset 127.5, exact in bf16) against constant ones in the B operand, so the accumulators start from the MFMA's inline zero instead of 96 loaded bias values
// per tile (-18 broadcast loads and ~100 register moves per wave-step).  l0t = this timestep's [3][64] fragments,
// written by k_prep_l0t, read with one coalesced 16-byte load per lane and tile.
template <int XSRC>
__device__ __forceinline__ void layer0_chain_t(const bf16x8* __restrict__ l0t, const float* x, f32x16 (&acc)[3], int lane) {
  const int h = lane >> 5;
  auto entry = [&](int j, bool feeder_is_upper) -> float {
    const bool owner_is_upper = (XSRC == 2);
    return feeder_is_upper == owner_is_upper ? x[j] : __shfl_xor(x[j], 32);
  };
  const bf16x8 w0 = l0t[lane], w1 = l0t[64 + lane], w2 = l0t[128 + lane];
  const float x8 = entry(8, true);
  float xe[8];
#pragma unroll
  for (int j = 0; j < 8; j++) xe[j] = entry(j, false);
  bf16x8 b;
#pragma unroll
  for (int j = 0; j < 8; j++) b[j] = (__bf16)(h ? (j == 0 ? x8 : (j < 4 ? 1.0f : 0.0f)) : xe[j]);  // slot 8h + j; 9..11 = ones
  acc[0] = mfma_bf16(w0, b, zero16<SO3X_PREC_BF16>());
  acc[1] = mfma_bf16(w1, b, zero16<SO3X_PREC_BF16>());
  acc[2] = mfma_bf16(w2, b, zero16<SO3X_PREC_BF16>());
}

// Layer 0, FULL variant: per-sample timestep; every lane evaluates the embedding
// entries of its own k-slots (the two lanes of a sample split the 56 sin/cos).
// The embedding k-steps are a real loop (not unrolled): 56 inlined sincos would
// otherwise be hoisted together and blow the register budget.
template <int PREC>
__device__ __forceinline__ void layer0_full(const char* __restrict__ wl, const float* x, int64_t t, const Freqs& fr,
                                            f32x16 (&acc)[3], int lane) {
  const int h = lane >> 5;
#pragma unroll
  for (int to = 0; to < 3; to++) acc[to] = zero16<PREC>();
  if constexpr (PREC == SO3X_PREC_F32) {
    const float* w = reinterpret_cast<const float*>(wl);
#pragma unroll
    for (int ks = 0; ks < 5; ks++) {  // slots 0..9: R and the constant one
      const float lo = x[2 * ks];
      const float hi = (2 * ks + 1 < 9) ? x[2 * ks + 1] : 1.0f;
      const float b = h ? hi : lo;
#pragma unroll
      for (int to = 0; to < 3; to++) acc[to] = mfma_f32(w[(to * 33 + ks) * 64 + lane], b, acc[to]);
    }
#pragma unroll 1
    for (int ks = 5; ks < 33; ks++) {  // slots 10..65: emb[2 ks + h - 10]
      const float b = emb_value(t, 2 * ks + h - 10, fr);
#pragma unroll
      for (int to = 0; to < 3; to++) acc[to] = mfma_f32(w[(to * 33 + ks) * 64 + lane], b, acc[to]);
    }
  } else {
    const bf16x8* w = reinterpret_cast<const bf16x8*>(wl);
    {
      bf16x8 b;  // slots 8h + j: R[0..7] | R[8], 1, 0...
#pragma unroll
      for (int j = 0; j < 8; j++) b[j] = (__bf16)(h ? (j == 0 ? x[8] : (j == 1 ? 1.0f : 0.0f)) : x[j]);
#pragma unroll
      for (int to = 0; to < 3; to++) acc[to] = mfma_bf16(w[(to * 5) * 64 + lane], b, acc[to]);
    }
#pragma unroll 1
    for (int ks = 1; ks < 5; ks++) {  // slots 16 ks + 8 h + j -> emb[16 (ks-1) + 8 h + j]
      bf16x8 b;
#pragma unroll
      for (int j = 0; j < 8; j++) {
        const int e = 16 * (ks - 1) + 8 * h + j;
        b[j] = (__bf16)(e < NEMB ? emb_value(t, e < NEMB ? e : 0, fr) : 0.0f);
      }
#pragma unroll
      for (int to = 0; to < 3; to++) acc[to] = mfma_bf16(w[(to * 5 + ks) * 64 + lane], b, acc[to]);
    }
  }
}

// The whole network on one 32-sample tile.  Returns the outputs of sample column
// (lane & 31) in v[0..NV-1]; only lanes of the LOWER half (h == 0) hold valid values.
template <int PREC, int VAR, int XSRC = 0, bool L0T = false, int NV = 3>
__device__ __forceinline__ void forward_tile(const char* __restrict__ img /*LDS weight image*/, const float* x,
                                             const float* __restrict__ beff, int64_t t, const Freqs* fr, float* v, int lane,
                                             const bf16x8* __restrict__ l0t = nullptr) {
  const int h = lane >> 5;
  constexpr int FB = frag_bytes<PREC>();
  f32x16 acc[3];
  Tile<PREC> cur;
  if constexpr (L0T) layer0_chain_t<XSRC>(l0t, x, acc, lane);
  else if constexpr (chain_layout(VAR)) layer0_chain<PREC, XSRC>(img, beff, x, acc, lane);
  else layer0_full<PREC>(img, x, t, *fr, acc, lane);
  constexpr bool FOLD = fold_scale<PREC, VAR>();
  const char* tab = img + (size_t)n_frags<PREC, VAR>() * FB;  // the SiLU table behind the fragments (FOLD only)
  activate<PREC, FOLD>(acc, cur, h, tab);
#pragma unroll
  for (int l = 1; l < 4; l++) {
    hidden_layer<PREC, 3>(img + (size_t)frag_hidden<PREC, VAR>(l) * FB, cur, acc, lane);
    activate<PREC, FOLD>(acc, cur, h, tab);
  }
  f32x16 last[1];
  hidden_layer<PREC, 1>(img + (size_t)frag_last<PREC, VAR>() * FB, cur, last, lane);
#pragma unroll
  for (int k = 0; k < NV; k++) v[k] = last[0][k];  // head outputs 0..5 = regs 0..5 of the lower half (head_of_row)
}

// ---- the two 32-sample tiles of a 64-sample wave through the network as ONE software-pipelined stream (bf16 chain) ------
// forward_tile runs a tile's 15 MFMAs of a layer, then its 33 table activations per lane: the matrix pipe idles through the
// activation and the vector port through most of the MFMAs (PMC, round 2: vector ALU busy 60 %, matrix pipe 34 %,
// waves parked 36 %).  A wave owns TWO independent tiles (samples 0..31 and 32..63 of its chunk), so the stages are skewed:
//     [MFMA A, layer l  ||  activation B, layer l-1]   [MFMA B, layer l  ||  activation A, layer l]   ...
// one tile's activation (116 vector instructions + 33 LDS lookups) sits in the shadow of the other tile's 15 MFMAs
// (15 x 32 cycles), inside one wave's instruction stream -- no reliance on a partner wave being in the complementary phase.
// Each stage is fenced (sched_barrier) and left to the scheduler inside; accumulators and operands of both tiles are
// live (96 + 40 registers), which the 8-wave workgroup's 256-register budget holds.
// `pre` = the five fragments of output tile 0, already in registers (fetched during the previous stage)
template <int NT, bool F16 = false>
__device__ __forceinline__ void mfma_layer_bf16(const char* __restrict__ wl, const Tile<SO3X_PREC_BF16>& in, f32x16 (&acc)[NT], int lane,
                                                const bf16x8 (&pre)[5]) {
  const bf16x8* w = reinterpret_cast<const bf16x8*>(wl);
#pragma unroll
  for (int to = 0; to < NT; to++) {
    f32x16 a = zero16<SO3X_PREC_BF16>();
#pragma unroll
    for (int ks = 0; ks < 5; ks++) a = mfma_op<F16>(to == 0 ? pre[ks] : w[(to * 5 + ks) * 64 + lane], in.b[ks], a);
    acc[to] = a;
  }
}
__device__ __forceinline__ void prefetch_tile0(const char* __restrict__ wl, int lane, bf16x8 (&pre)[5]) {
  const bf16x8* w = reinterpret_cast<const bf16x8*>(wl);
#pragma unroll
  for (int ks = 0; ks < 5; ks++) pre[ks] = w[ks * 64 + lane];
}
// the layer-0 B operand of tile A (XSRC 1: column c lives in lane c) or B (XSRC 2: lane 32 + c), as layer0_chain_t builds it
template <int XSRC>
__device__ __forceinline__ bf16x8 l0_operand(const float* x, int lane) {
  const int h = lane >> 5;
  auto entry = [&](int j, bool feeder_is_upper) -> float {
    const bool owner_is_upper = (XSRC == 2);
    return feeder_is_upper == owner_is_upper ? x[j] : __shfl_xor(x[j], 32);
  };
  const float x8 = entry(8, true);
  float xe[8];
#pragma unroll
  for (int j = 0; j < 8; j++) xe[j] = entry(j, false);
  bf16x8 b;
#pragma unroll
  for (int j = 0; j < 8; j++) b[j] = (__bf16)(h ? (j == 0 ? x8 : (j < 4 ? 1.0f : 0.0f)) : xe[j]);  // slot 8h + j; 9..11 = ones
  return b;
}
// Both layer-0 B operands of a 64-sample wave at once (tile A = the lower lane half's samples, tile B = the upper half's), with
// v_permlane32_swap_b32 (gfx950: swaps the upper 32 lanes of its first operand with the lower 32 of its second) instead of nine
// ds_bpermute exchanges and two select chains.  Every lane packs its OWN entries -- P_w = (x[2w], x[2w+1]), Q = (x[8], 1) -- and
// word w of the two operands is one swap:   swap(D = P_w, S = C_w)  ->  D' = [P_w of the lower half | C_w of the lower half] = tile A,
//                                                                   S' = [P_w of the upper half | C_w of the upper half] = tile B,
// with C_0 = Q (slot 8 = R[8], slot 9 = one) and C_1 = (1, 1), C_2 = C_3 = 0 (slots 10, 11 = ones, 12..15 = padding).
// Same bf16 bits as l0_operand<1> / <2>.
template <bool F16 = false>
__device__ __forceinline__ void l0_operands_pair(const float* x, bf16x8& bA, bf16x8& bB) {
  typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
  auto pk = [](float a, float b) { return pack_pair<F16>(a, b); };
  const uint32_t c[4] = {pk(x[8], 1.0f), F16 ? 0x3C003C00u : 0x3F803F80u, 0u, 0u};   // (1, 1) as two halves / two bf16
  uint32_t wa[4], wb[4];
#pragma unroll
  for (int w = 0; w < 4; w++) {
    const auto r = __builtin_amdgcn_permlane32_swap(pk(x[2 * w], x[2 * w + 1]), c[w], false, false);
    wa[w] = r[0]; wb[w] = r[1];
  }
  bA = __builtin_bit_cast(bf16x8, u32x4_t{wa[0], wa[1], wa[2], wa[3]});
  bB = __builtin_bit_cast(bf16x8, u32x4_t{wb[0], wb[1], wb[2], wb[3]});
}
// The order WITHIN a stage.  An MFMA holds the SIMD's vector issue for 8 of its 32 cycles; about six 4-cycle instructions
// fit the rest (MI355X_MICROARCH.md, issue-cost rows).  The compiler's own order put the lookups between the MFMAs and left
// the ~50 multiply-adds and packings of a stage as a tail behind them, with the matrix pipe idle (sched_group_barrier
// pipelines did not move it); stage_gaps below lays the stage out gap by gap instead.
// One stage of the paired stream laid out gap by gap: the 15 MFMAs of tile X's layer, and in the gap behind each of them a
// slice of tile Y's activation -- table lookups for about three values, the multiply-adds of the lookups issued two gaps
// earlier, the packing of finished pairs -- and ONE weight-fragment read (five MFMAs ahead: the fragments of output tiles 1
// and 2, then the next stage's first five into `ring`).  A sched_barrier closes every gap, so the emitted order is this one.
template <bool WIDE, bool F16 = false>
__device__ __forceinline__ void stage_gaps(const char* __restrict__ wl, const char* __restrict__ wnext, const Tile<SO3X_PREC_BF16>& inX,
                                           f32x16 (&accX)[3], const f32x16 (&accY)[3], Tile<SO3X_PREC_BF16>& curY, bf16x8 (&ring)[5],
                                           int lane, int h, const char* tab, uint32_t lt) {
  const bf16x8* w = reinterpret_cast<const bf16x8*>(wl);
  const bf16x8* wn = reinterpret_cast<const bf16x8*>(wnext);
  float2 e[33];
  float val[33];
  uint32_t pk[16];
  constexpr int LAG = 2;  // gaps between a lookup and its multiply-add (the LDS round trip under load)
  constexpr int LAST_LOOKUP_GAP = 13 - LAG;  // lookups in gaps 0..13-LAG, multiply-adds LAG gaps behind, packing one more
  auto first_of = [](int g) { return g <= 0 ? 0 : (g > LAST_LOOKUP_GAP ? 33 : (33 * g) / (LAST_LOOKUP_GAP + 1)); };
#pragma unroll
  for (int g = 0; g < 15; g++) {
    const int to = g / 5, ks = g % 5;
    accX[to] = mfma_op<F16>(ring[ks], inX.b[ks], ks == 0 ? zero16<SO3X_PREC_BF16>() : accX[to]);
    ring[ks] = g < 10 ? w[(g + 5) * 64 + lane] : wn[(g - 10) * 64 + lane];
#pragma unroll
    for (int q = first_of(g); q < first_of(g + 1); q++) {
      const float u = q < 32 ? accY[q >> 4][q & 15] : accY[2][0];
      if constexpr (WIDE) {
        e[q] = wide_tab_entry(u, lt);
      } else {
        const unsigned idx = __builtin_amdgcn_cvt_pk_u8_f32(u, 0u, 0u);
        e[q] = *reinterpret_cast<const float2*>(tab + idx * 8);
      }
    }
#pragma unroll
    for (int q = first_of(g - LAG + 1) - 1; q >= first_of(g - LAG); q--)  // newest lookup first: ONE counted wait covers the gap's multiply-adds
      val[q] = fmaf(e[q].y, q < 32 ? accY[q >> 4][q & 15] : accY[2][0], e[q].x);
#pragma unroll
    for (int j = 0; j < 16; j++) {
      const bool ready_now = 2 * j + 1 < first_of(g - LAG), ready_before = 2 * j + 1 < first_of(g - LAG - 1);
      if (ready_now && !ready_before) {
        pk[j] = pack_pair<F16>(val[2 * j], val[2 * j + 1]);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
#pragma unroll
  for (int i = 0; i < 4; i++) curY.b[i] = __builtin_bit_cast(bf16x8, u32x4_t{pk[4 * i], pk[4 * i + 1], pk[4 * i + 2], pk[4 * i + 3]});
  const float last8[8] = {h ? 1.0f : val[32], h ? 1.0f : 0.0f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};  // row 64 | the constant ones of rows 68, 69
  curY.b[4] = pack_octet<F16>(last8);
}

template <bool WIDE = false, bool F16 = false>
__device__ __forceinline__ void forward_pair_bf16(const char* __restrict__ img, const float* x, bf16x8 (&l0w)[3],
                                                  float* va, float* vb, int lane, uint32_t lt = 0,
                                                  const bf16x8* __restrict__ l0next = nullptr) {
  constexpr int PREC = SO3X_PREC_BF16, VAR = CHAIN, FB = frag_bytes<PREC>();
  const int h = lane >> 5;
  const char* tab = img + (size_t)n_frags<PREC, VAR>() * FB;
  f32x16 accA[3], accB[3];
  Tile<PREC> curA, curB;
  {  // layer 0 of both tiles from this timestep's three A fragments
    const bf16x8 w0 = l0w[0], w1 = l0w[1], w2 = l0w[2];
    bf16x8 bA, bB;
    l0_operands_pair<F16>(x, bA, bB);
    accA[0] = mfma_op<F16>(w0, bA, zero16<PREC>()); accA[1] = mfma_op<F16>(w1, bA, zero16<PREC>()); accA[2] = mfma_op<F16>(w2, bA, zero16<PREC>());
    accB[0] = mfma_op<F16>(w0, bB, zero16<PREC>()); accB[1] = mfma_op<F16>(w1, bB, zero16<PREC>()); accB[2] = mfma_op<F16>(w2, bB, zero16<PREC>());
  }
  // the NEXT step's three layer-0 fragments go straight into the registers this step's six MFMAs have just read (an L2 round
  // trip that lands during the stages; no second set of twelve registers, no twelve moves per step)
  if (l0next) {
#pragma unroll
    for (int k = 0; k < 3; k++) l0w[k] = l0next[64 * k + lane];
  }
  // every stage also fetches the first five weight fragments of the NEXT stage (`pre`): a stage's MFMA chain starts on
  // registers instead of waiting ~120 cycles for its first LDS reads behind the fence
  bf16x8 pre[5];
  prefetch_tile0(img + (size_t)frag_hidden<PREC, VAR>(1) * FB, lane, pre);
  activate_bf16<true, WIDE, F16>(accA, curA, h, tab, lt);
  const char* wlast = img + (size_t)frag_last<PREC, VAR>() * FB;
  // Wave priority by phase (s_setprio): the six MFMA stages at 3, the output-layer stages
  // at 1, everything else of a step -- its top, layer 0, the all-vector reverse step -- at 0.  The two waves of a SIMD run
  // the same program at equal priority and the arbiter interleaved them instruction by instruction; with the stages on top a
  // wave in its stages keeps the matrix pipe fed and its partner's vector work fills what is left: 5.97 -> 5.51 ms per 100
  // steps.  Any split helped (reverse step on top: 5.82; network on top: 5.78; a fixed winner per SIMD: nothing).
  __builtin_amdgcn_s_setprio(3);
#pragma unroll
  for (int l = 1; l < 4; l++) {
    const char* wl = img + (size_t)frag_hidden<PREC, VAR>(l) * FB;
    const char* wnext = l < 3 ? img + (size_t)frag_hidden<PREC, VAR>(l + 1) * FB : wlast;
    __builtin_amdgcn_sched_barrier(0);
    stage_gaps<WIDE, F16>(wl, wl, curA, accA, accB, curB, pre, lane, h, tab, lt);     // MFMA A, layer l || activation B, layer l-1
    stage_gaps<WIDE, F16>(wl, wnext, curB, accB, accA, curA, pre, lane, h, tab, lt);  // MFMA B, layer l || activation A, layer l
  }
  __builtin_amdgcn_s_setprio(1);
  f32x16 lastA[1], lastB[1];
  __builtin_amdgcn_sched_barrier(0);
  mfma_layer_bf16<1, F16>(wlast, curA, lastA, lane, pre);  // head A             ||
  activate_bf16<true, WIDE, F16>(accB, curB, h, tab, lt);            // activation B, layer 3 (head B below takes the SAME five fragments)
  __builtin_amdgcn_sched_barrier(0);
  mfma_layer_bf16<1, F16>(wlast, curB, lastB, lane, pre);
  __builtin_amdgcn_s_setprio(0);
#pragma unroll
  for (int k = 0; k < 3; k++) { va[k] = lastA[0][k]; vb[k] = lastB[0][k]; }
}

// cooperative copy of the weight image (global workspace -> LDS), 16 B per lane
__device__ __forceinline__ void load_image(const void* __restrict__ gimg, char* lds, int bytes) {
  const float4* s = reinterpret_cast<const float4*>(gimg);
  float4* d = reinterpret_cast<float4*>(lds);
  for (int i = threadIdx.x; i < bytes / 16; i += blockDim.x) d[i] = s[i];
}

// ---- host-side launch helpers implemented in so3x_mlp.hip -------------------------
size_t image_bytes_rt(int precision, int variant);
// writes the weight image at workspace[0 .. image) and, for CHAIN with T > 0, the
// effective-bias table beff[T][96] right after it (16-B aligned).
// One launch: [the weight image] [the transposed image -> wt, optional] [the per-timestep tables when T > 0].
// want_image = false skips the forward image (the backward with a stash never reads it).
// zero_word (optional): a device word the launch clears (the arrival ticket of a kernel that follows in the stream).
// t_count > 0: only the per-timestep rows t_first .. t_first + t_count - 1 are built (a chain launch of a few steps reads no others).
// f16: the bf16 images' 16-bit slots are filled with IEEE half bits instead (the chain kernel's f16-operand leg)
int launch_prep(hipStream_t s, const float* params, int precision, int variant, int T, void* workspace, int nout = 3,
                void* wt = nullptr, bool want_image = true, unsigned* zero_word = nullptr, int t_first = 0, int t_count = 0, bool f16 = false);
int launch_prep_l0t(hipStream_t s, const float* params, int T, void* workspace, int t_first = 0, int t_count = 0, bool f16 = false);
size_t beff_offset(int precision, int variant);
// tables that follow the image for chain-layout variants: beff [T][96] fp32, then emb [T][56] fp32
inline size_t emb_offset(int precision, int variant, int T) { return beff_offset(precision, variant) + (size_t)T * 96 * sizeof(float); }
// bf16 CHAIN only: per-timestep layer-0 fragments [T][3][64][8 bf16] after the beff table (16-byte aligned: 384 B rows)
inline size_t l0t_offset(int T) { return emb_offset(SO3X_PREC_BF16, CHAIN, T); }
inline size_t l0t_end(int T) { return l0t_offset(T) + (size_t)T * 3 * 1024; }
// ... then h0 [T][96] bf16: the layer-0 input row of timestep t as the backward's LDS image wants it (slot s of the 96 =
// input slot s: 10..65 the embedding, zeros elsewhere), so a lane fetches its 48 slots with six 16-byte loads
inline size_t h0_offset(int precision, int variant, int T) { return emb_offset(precision, variant, T) + (size_t)T * NEMB * sizeof(float); }
inline size_t tables_end(int precision, int variant, int T) { return h0_offset(precision, variant, T) + (size_t)T * 96 * 2; }
const Freqs& host_freqs();

}  // namespace mlp
}  // namespace so3x
