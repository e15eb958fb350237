// so3x_train.hpp -- device helpers shared by the kernels of a training step of the 65-wide score network
// (so3x_mlp_bwd.hip: the staged step = training forward with its pre-activation stash + fused backward;
//  so3x_train_fused.hip: noising + forward + backward as ONE kernel): the LDS hand-over images between the chain waves and
// the dW waves, the dH MFMAs on packed operands, the ownership of the 39 dW tiles, the MSE epilogue's arrival ticket and the
// workspace layout of a step.
#pragma once
#include "so3x_common.hpp"
#include "so3x_mlp.hpp"

namespace so3x {
namespace train {
using namespace so3x::mlp;

constexpr int DW_BLOCKS = 256;                  // workgroups (= partial dW slabs) of a backward launch: one per CU
template <int PREC> __host__ __device__ constexpr int wt_bytes() { return wt_nfrags<PREC>() * frag_bytes<PREC>(); }


constexpr int FIMG_COLS = 192;                 // dZ block [0, 96), H block [96, 192)
constexpr int FIMG_PITCH = FIMG_COLS * 2;      // bytes per sample row
constexpr int FIMG_BYTES = 32 * FIMG_PITCH;    // 12,288 per (wave, layer)

__device__ __forceinline__ int fimg_off(int row, int col /*multiple of 4*/) {
  int ch = col >> 2;
  ch ^= (row & 7) | ((((row >> 1) ^ (row >> 3)) & 1) << 3);
  return row * FIMG_PITCH + ch * 8;
}
// Addressing is split into a per-lane part computed once per round (a handful of VGPRs) and compile-time
// constants that fold into the DS instructions' offset field; the round loop makes the per-lane parts opaque
// (empty asm) so the compiler does not hoist ~800 loop-invariant address registers and spill them.
struct FimgStoreLane { int rowbase, swz8; };  // rowbase = row * pitch; swz8 = 8 * swizzle(row)
__device__ __forceinline__ FimgStoreLane fimg_store_lane(int row) {
  return FimgStoreLane{row * FIMG_PITCH, 8 * ((row & 7) | ((((row >> 1) ^ (row >> 3)) & 1) << 3))};
}
// chunk index `ch` (= column / 4) is a compile-time constant plus the lane-half bit h in bit 0
__device__ __forceinline__ void fimg_store4(char* img, const FimgStoreLane& L, int ch, float a, float b, float c, float d) {
  typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
  bf16x4 v = {(__bf16)a, (__bf16)b, (__bf16)c, (__bf16)d};
  *reinterpret_cast<bf16x4*>(img + L.rowbase + ((ch * 8) ^ L.swz8)) = v;
}
// transposed-read lane offsets: [part (rows +0 / +4)][cbit (bit 3 of the tile's first chunk)]
struct FimgReadLane { int off[2][2]; };
__device__ __forceinline__ FimgReadLane fimg_read_lane(int lane) {
  const int h = lane >> 5, l32 = lane & 31, G = l32 >> 4, q = (l32 & 15) >> 2, pp = l32 & 3;
  FimgReadLane L;
#pragma unroll
  for (int part = 0; part < 2; part++)
#pragma unroll
    for (int cbit = 0; cbit < 2; cbit++) {
      const int row = 8 * h + q + 4 * part;                                  // row within a 16-sample k-step
      const int sw = ((q + 4 * part) & 7) | ((((q >> 1) ^ h) & 1) << 3);       // swizzle(16 ks + row): ks drops out
      L.off[part][cbit] = row * FIMG_PITCH + 8 * ((8 * cbit + 4 * G + pp) ^ sw);
    }
  return L;
}
// MFMA operand (8 bf16 = samples 16 ks + 8 h + 0..7 of feature cb + (lane & 31)); cb in {0,32,64,96,128,160}
__device__ __forceinline__ bf16x8 fimg_frag(const char* img, const FimgReadLane& L, int cb, int ks) {
  typedef short s16x4 __attribute__((ext_vector_type(4)));
  typedef __attribute__((address_space(3))) s16x4* lds_p;
  const int chb = cb >> 2, cbit = (chb >> 3) & 1;
  const int konst = 16 * ks * FIMG_PITCH + (chb & ~15) * 8;                  // folds into the instruction offset
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(img + konst + L.off[0][cbit]));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(img + konst + L.off[1][cbit]));
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, v);
}


__device__ __forceinline__ uint32_t pack_bf16x2(float a, float b) {
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  bf16x2 v = {(__bf16)a, (__bf16)b};
  return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ void fimg_store_pk(char* img, const FimgStoreLane& L, int ch, uint32_t lo, uint32_t hi) {
  *reinterpret_cast<uint2*>(img + L.rowbase + ((ch * 8) ^ L.swz8)) = uint2{lo, hi};
}

// dH = W^T dZ with dZ given as packed bf16 pairs (the MFMA operand bits as they are)
template <int PREC, int L>
__device__ __forceinline__ void dh_layer_pk(const void* __restrict__ wt, const uint32_t (&pdz)[17], f32x16 (&dh)[3], int lane) {
  const bf16x8* w = reinterpret_cast<const bf16x8*>(wt);
  constexpr int KS = L < 4 ? 5 : 1;
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  bf16x8 b[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ks++) {
    const u32x4 v = ks < 4 ? u32x4{pdz[4 * ks], pdz[4 * ks + 1], pdz[4 * ks + 2], pdz[4 * ks + 3]} : u32x4{pdz[16], 0u, 0u, 0u};
    b[ks] = __builtin_bit_cast(bf16x8, v);
  }
#pragma unroll
  for (int to = 0; to < 3; to++) {
    __builtin_amdgcn_sched_barrier(0);
    f32x16 a = zero16<PREC>();
#pragma unroll
    for (int ks = 0; ks < KS; ks++) {
      a = mfma_bf16(w[(size_t)wt_frag<PREC>(L, to, ks) * 64 + lane], b[ks], a);
    }
    dh[to] = a;
  }
}


__host__ __device__ constexpr int dw_row(int dwi, int l) { return (dwi - l) & 3; }              // 0..2 = the row, 3 = idle in layer l
__host__ __device__ constexpr int dw_slot(int dwi, int l) {                                       // index of layer l among the wave's active layers
  int c = 0;
  for (int q = 0; q < l; q++) c += dw_row(dwi, q) != 3;
  return c;
}


// Arrival ticket of a grid (MI355X_MICROARCH.md, hand-off table row 1): the calling thread has stored this block's
// contribution with agent-scope (sc1) stores; it drains them, takes a ticket, and the block whose ticket is the last one
// may read every block's contribution with agent-scope loads.  The last arriver resets the ticket, so the word is zero
// again when the launch ends; the prep launch of the step clears it anyway (a fresh workspace holds garbage).  One
// calling thread per block.
__device__ __forceinline__ bool last_block_arrives(unsigned* ticket) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned mine = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (mine != gridDim.x - 1) return false;
  __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return true;
}

// What the fused MSE epilogue of the training forward needs (diffusion.py:357: F.mse_loss over the n x 3 outputs).
struct LossArgs {
  const float* target;   // [n][3] regression targets
  float* dout;           // [n][3] d loss / d out = 2 (out - target) / (3 n)
  float* loss;           // [1]
  float* status;         // [1] inside the step's workspace: a copy of the loss for the kernels that consume the step's slabs -- a
                         // non-finite value (a timed-out hand-shake) makes so3x_train_bwd_reduce{,_adam} leave everything untouched
  double* partial;       // [gridDim.x] per-block sums of squared differences
  unsigned* ticket;      // arrival ticket, zero between launches
  int64_t* rng_counter;  // optional: device-resident Philox offset of the noise draw, incremented once per step
  float dscale;          // 2 / (3 n)
  double inv_count;      // 1 / (3 n)
};


struct NoiseArgs {
  const float* sched; const float* trap_q; const uint16_t* guide_q; const float* x0; const int64_t* t; int64_t* t_draw;
  const float* axes; const float* unif; const int64_t* rng_offset_dev; uint64_t seed, rng_offset; int64_t index_base; int T, quirk_col0;
};

// Training-step workspace (so3x_train_fwd / so3x_train_bwd): what ONE prep launch builds for both halves of the step,
// the dW partial slabs, the regression targets and the loss bookkeeping.
struct TrainLayout { size_t wt, slabs, target, partial, ticket, end; };
inline TrainLayout train_layout(int64_t n, int T) {
  constexpr int PREC = SO3X_PREC_BF16;
  TrainLayout L;
  L.wt = (tables_end(PREC, GATHER, T) + 255) & ~(size_t)255;
  L.slabs = (L.wt + (size_t)wt_nfrags<PREC>() * frag_bytes<PREC>() + 255) & ~(size_t)255;
  L.target = L.slabs + (size_t)DW_BLOCKS * NPARAMS_MAX * sizeof(float);
  L.partial = (L.target + (size_t)(n > 0 ? n : 0) * 3 * sizeof(float) + 255) & ~(size_t)255;
  L.ticket = L.partial + 512 * sizeof(double);
  L.end = L.ticket + 256;   // (the ticket word at + 0, the step's status word -- LossArgs::status -- at + 64)
  return L;
}


}  // namespace train
}  // namespace so3x
