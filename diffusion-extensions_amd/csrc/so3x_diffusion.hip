// so3x_diffusion.hip -- fused SO3Diffusion steps:
//   q_sample + training target (diffusion.py:339-355), reverse mean (291-313) and the
//   chain-resident reverse sampler p_sample / p_sample_loop (315-337) with the score
//   network on the matrix cores.
#include <stdlib.h>
#include <string.h>
#include "so3x_common.hpp"
#include "so3x_math.hpp"
#include "so3x_igso3.hpp"
#include "so3x_mlp.hpp"
#include "so3x_reverse_step.hpp"

using namespace so3x;
using namespace so3x::mlp;

// (Round 4: the forms this file carried behind macros -- the record's DMA issued late, the builtin instead of the inline-assembly
//  LDS-DMA, a second register set for the next step's layer-0 fragments, a per-SIMD stage token, phase stamps, ablations -- were
//  measured in rounds 2 and 3 (profiles/r02_ab_chain_*.json, r03_ab_chain_*.json), lost or served their purpose, and are gone
//  from the tree; what is here is the shipped kernel.  The environment-switched forms of the A/B library stay: SO3X_AB_BUILD.)
// The step's five LDS-DMA instructions are inline assembly, so that the compiler's wait-count pass does not see an LDS writer in
// flight and put `s_waitcnt vmcnt(0)` in front of the next LDS read (the weight fragments of layer 0: an exposed L2 round trip at
// the top of every step); the kernel's own `s_waitcnt vmcnt(0)` in front of the reverse step orders the record's arrival
// against its reads.  The next step's layer-0 fragments are loaded into the registers the current step's layer 0 has just read.
namespace {

// ---------------------------------------------------------------------------------------
// A12: noise draw + forward noising + regression target, one pass, 84 B/sample algorithmic
// (36 x0 in, 36 x_t + 12 target out; +8 for t).
// ---------------------------------------------------------------------------------------
// Wave-private staging (one 64-sample tile per wave, no workgroup barrier), launched one tile per wave on an
// oversubscribed grid like k_logprob_score: the per-sample CDF-row search is a chain of ~10 dependent L2 loads, and
// only other resident waves hide it.
// t == nullptr: the timesteps are not given but drawn here -- t_i = floor(T * w_i / 2^32) from the fourth word of the
// sample's Philox block (the other three feed the noise); SO3Diffusion.forward's `t = randint(0, T, (b,))` (diffusion.py:373)
// keyed, like the noise, by the global sample index.  Given timesteps are clamped to [0, T-1] (the reference raises IndexError;
// a kernel cannot, and must not read outside its tables).  t_draw (optional): where the timesteps actually used are written
// for the kernels that follow.
// LEAN: the training step's form (noise drawn here: no noise_in, no explicit axes / uniforms) as its own instantiation at
// 64 registers -- eight waves per SIMD hold all 8,192 tiles of a 2^19-sample step at once; at the generic kernel's 67
// registers (seven waves) the last eighth of the tiles ran as a second round: 23.7 -> 20.1 us (tools/ab/ab_qsample.py).
template <bool LEAN>
__global__ void __launch_bounds__(kBlock, LEAN ? 8 : 6)
k_q_sample_target(const float* __restrict__ sched, int T, const float* __restrict__ trap_q,
                  const uint16_t* __restrict__ guide_q, const float* __restrict__ x0,
                  const int64_t* __restrict__ t, int64_t* __restrict__ t_draw, int quirk_col0, const float* __restrict__ noise_in_,
                  const float* __restrict__ axes_, const float* __restrict__ unif_, uint64_t seed, uint64_t rng_offset,
                  const int64_t* __restrict__ rng_offset_dev, int64_t index_base, float* __restrict__ x_t,
                  float* __restrict__ target, float* __restrict__ noise_out, int64_t n) {
  __shared__ __attribute__((aligned(16))) float sm[kBlock / kWave][kWave * 9];
  const float* noise_in = LEAN ? nullptr : noise_in_;
  const float* axes = LEAN ? nullptr : axes_;
  const float* unif = LEAN ? nullptr : unif_;
  if (rng_offset_dev) rng_offset += (uint64_t)rng_offset_dev[0];  // device-resident part of the counter (hipGraph replays)
  float* wl = sm[threadIdx.x >> 6];
  const int lane = threadIdx.x & 63;
  const int64_t ntiles = (n + kWave - 1) / kWave;
  const int64_t wave = (int64_t)blockIdx.x * (kBlock / kWave) + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * (kBlock / kWave);
  auto drawn_t = [&](uint32_t w) -> int64_t { return (int64_t)(((uint64_t)w * (uint64_t)T) >> 32); };
  auto clamp_t = [&](int64_t v) -> int64_t { return v < 0 ? 0 : (v >= T ? T - 1 : v); };  // as k_p_mean: never index outside a table
  // distributions.py:42-43: column 0 == sample 0's eps.  With drawn timesteps that is GLOBAL sample 0's draw (Philox index 0,
  // whatever this shard's index_base), so every shard of a data-parallel run uses the row the single-process run uses; with
  // caller-supplied timesteps it is this call's t[0], as in the reference (the caller decides what a shard's t[0] is).
  const int64_t wrow_t = !quirk_col0 ? -1 : (t ? clamp_t(t[0]) : drawn_t(philox4x32_10(seed, (uint64_t)0, rng_offset).w));
  for (int64_t tile = wave; tile < ntiles; tile += nwaves) {
    const int64_t base = tile * kWave;
    const int cnt = (int)((n - base) < kWave ? (n - base) : kWave);
    const int64_t idx = base + lane;
    const bool live = lane < cnt;
    // LEAN: the tile's x0 is requested FIRST (LDS-DMA straight into the wave's staging buffer: no registers held) and lands while the noise chain runs
    // -- Philox, timestep, guide, CDF window: two dependent L2 round trips -- instead of behind it: with one tile per wave and
    // every wave of the launch resident at once, all waves walk their dependent round trips in step and each one adds to the
    // kernel's time
    const bool prefetched = LEAN && cnt == kWave && ((reinterpret_cast<uintptr_t>(x0 + base * 9) & 15) == 0);
    if (prefetched) wave_dma9(x0, base, wl);
    Philox4 r;
    if (!t || (!noise_in && !axes)) r = philox4x32_10(seed, (uint64_t)(index_base + idx), rng_offset);
    int64_t tt;
    if (t) tt = clamp_t(t[live ? idx : base]);
    else tt = drawn_t(r.w);
    if (t_draw && live) t_draw[idx] = tt;  // the timesteps the kernels behind this one gather with: drawn here, or the caller's, clamped
    float nz[9];
    float nlog[3] = {0.f, 0.f, 0.f};  // LEAN: axis * angle of the noise drawn here
    if (noise_in) {
      wave_load_rows<9>(noise_in, base, cnt, wl, nz);
    } else {
      float ax[3], u;
      if (axes) {
        float a[3];
        wave_load_rows<3>(axes, base, cnt, wl, a);
        float nrm = sqrtf(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]);       // distributions.py:36
        ax[0] = a[0] / nrm; ax[1] = a[1] / nrm; ax[2] = a[2] / nrm;
        float n2 = sqrtf(ax[0] * ax[0] + ax[1] * ax[1] + ax[2] * ax[2]);  // util.py:201
        ax[0] /= n2; ax[1] /= n2; ax[2] /= n2;
        u = live ? unif[idx] : 0.5f;
      } else {
        unit_axis(r.x, r.y, ax);
        u = u01(r.z);
      }
      const int64_t tq = tt;
      const float* row = trap_q + tq * 999;
      const float* wrow = wrow_t >= 0 ? trap_q + wrow_t * 999 : row;
      const float ang = igso3_angle_global(row, wrow, SO3X_KNOTS_DATA, u, guide_q ? guide_q + tq * kGuidePitch : nullptr);
      exp_axis_angle(ax, ang, nz);
      if constexpr (LEAN) { nlog[0] = ax[0] * ang; nlog[1] = ax[1] * ang; nlog[2] = ax[2] * ang; }
    }
    float x[9], w[3], xs[9], xt[9];
    if (prefetched) wave_dma9_commit(wl, x);
    else wave_load_rows<9>(x0, base, cnt, wl, x);
    const float k = sched[S_SQRT_AC * T + tt];
    log3(x, w);
    w[0] *= k; w[1] *= k; w[2] *= k;
    exp3(w, xs);                      // so3_scale(x_start, sqrt(abar_t)), diffusion.py:344-345
    mul33(xs, nz, xt);                // x_blend @ noise, :346
    if (x_t) wave_store_rows<9>(x_t, base, cnt, wl, xt);
    if (target) {
      float lw[3];
      // skew2vec(log_rmat(noise)) * (1/eps), :355.  LEAN (the noise was built here from a unit axis and an angle in [0, pi]):
      // log(exp(hat(axis * angle))) IS axis * angle -- taken directly instead of through the matrix and an atan2 log, which
      // can only return it with the fp32 rounding of nine matrix entries on top (amplified by 1 / (pi - angle) near pi)
      if constexpr (LEAN) { lw[0] = nlog[0]; lw[1] = nlog[1]; lw[2] = nlog[2]; }
      else log3(nz, lw);
      const float ie = 1.0f / sched[S_SQRT_1MAC * T + tt];
      float tg[3] = {lw[0] * ie, lw[1] * ie, lw[2] * ie};
      wave_store_rows<3>(target, base, cnt, wl, tg);
    }
    if (noise_out) wave_store_rows<9>(noise_out, base, cnt, wl, nz);
  }
}

// tvec == nullptr: one shared timestep t (p_sample, diffusion.py:315-326); else per-sample timesteps tvec[i * t_stride], as the
// reference's extract(coef, t, shape) gathers them (diffusion.py:291-306)
__global__ void __launch_bounds__(kBlock)
k_p_mean(const float* __restrict__ sched, int T, const float* __restrict__ x, const float* __restrict__ v, int t,
         const int64_t* __restrict__ tvec, int64_t t_stride, float* __restrict__ x0hat, float* __restrict__ mean, int64_t n) {
  __shared__ __attribute__((aligned(16))) float sm[kTile * 9];
  const int64_t ntiles = (n + kTile - 1) / kTile;
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t base = tile * kTile;
    const int cnt = (int)((n - base) < kTile ? (n - base) : kTile);
    int ts = t;
    if (tvec) {
      const int64_t i = base + ((int)threadIdx.x < cnt ? threadIdx.x : 0);
      const int64_t tv = tvec[i * t_stride];
      ts = (int)(tv < 0 ? 0 : (tv >= T ? T - 1 : tv));
    }
    const float a = sched[S_RECIP * T + ts], b = sched[S_RECIPM1 * T + ts], c1 = sched[S_COEF1 * T + ts], c2 = sched[S_COEF2 * T + ts];
    float xx[9], vv[3], xh[9], mm[9];
    load_rows<9>(x, base, cnt, sm, xx);
    load_rows<3>(v, base, cnt, sm, vv);
    p_mean_one(xx, vv, a, b, c1, c2, xh, mm);
    if (x0hat) store_rows<9>(x0hat, base, cnt, sm, xh);
    store_rows<9>(mean, base, cnt, sm, mm);
  }
}

// ---- per-timestep CDF records of the bf16 chain kernel: [999 floats of trap_p row t][258 uint16 of guide row t][pad] = 4,608
//      bytes, 16-byte aligned, so that a wave brings its step's row into LDS with five LDS-DMA instructions at the top of the
//      step and the inverse-CDF search of the reverse step (a chain of 4-8 dependent loads) runs on LDS instead of L2.
constexpr int kCdfRec = 4608, kCdfGuideOff = 999 * 4;
static_assert(kCdfGuideOff + kGuidePitch * 2 <= kCdfRec && kCdfRec % 512 == 0, "record layout");
inline size_t cdf_offset(int T) { return (l0t_end(T) + 255) & ~(size_t)255; }
inline size_t cdf_end(int T) { return cdf_offset(T) + (size_t)T * kCdfRec; }
// ... then two 64-bit words the chain kernel leaves per launch: the shader-clock ticks (s_memtime) and the 100 MHz reference ticks
// (s_memrealtime) wave 0 of workgroup 0 spent in it -- their quotient x 100 MHz is the clock the launch actually ran at
// (MI355X_MICROARCH.md, DVFS give-back), which bench.py needs to price the vector issue port.  Two scalar instructions at
// either end of a launch, none inside the step loop.
constexpr size_t kClockBytes = 64;
inline size_t clock_offset_bf16(int T) { return (cdf_end(T) + 63) & ~(size_t)63; }
inline size_t clock_offset_f32(int T) { return (beff_offset(SO3X_PREC_F32, CHAIN) + (size_t)T * 96 * sizeof(float) + 63) & ~(size_t)63; }
__global__ void __launch_bounds__(256) k_prep_cdf(const float* __restrict__ trap_p, const uint16_t* __restrict__ guide_p, char* __restrict__ rec,
                                                  int t_first) {
  const int t = t_first + blockIdx.x;
  char* r = rec + (size_t)t * kCdfRec;
  for (int i = threadIdx.x; i < 999; i += 256) reinterpret_cast<float*>(r)[i] = trap_p[(size_t)t * 999 + i];
  for (int i = threadIdx.x; i < (kCdfRec - kCdfGuideOff) / 2; i += 256)
    reinterpret_cast<uint16_t*>(r + kCdfGuideOff)[i] = (guide_p && i < kGuidePitch) ? guide_p[(size_t)t * kGuidePitch + i] : (uint16_t)0;
}
constexpr int kChainKnotsBytes = 4096, kChainRowBufBytes = 8 * kCdfRec;  // LDS behind the weight image (WIDE variant)

// ---------------------------------------------------------------------------------------
// A13: chain-resident reverse sampler.
//  * one wave owns 64 samples; each lane keeps its rotation in 9 VGPRs for ALL n_steps
//    steps (HBM traffic for a whole chain: 72 B/sample);
//  * the score network runs on the matrix cores as two 32-sample tiles per wave, weights
//    LDS-resident (so3x_mlp.hpp); the time embedding enters as a per-timestep effective
//    bias row (prep kernel), so the per-step reads are: 96 floats of bias, 5 schedule
//    scalars, the 4 KB CDF row of sigma_t (L1/L2-resident, shared by every wave);
//  * waves never synchronise with each other after the weight image is loaded;
//  * noise: Philox keyed (seed; global sample index, rng_offset + t) or explicit draws.
// ---------------------------------------------------------------------------------------
// FAST: hardware sine / cosine in the reverse step (bf16 default; so3x_math.hpp).
// bf16: ONE workgroup of 8 waves per CU (the image is 55 KB and the lane-replicated SiLU table 64 KB: 119 of the CU's 160 KB).  Measured at B = 2^20 (profiles/r02_ab_chain_blocks.json): 8 waves 6.96 ms per 100 steps, 12 waves
// 7.04, two workgroups of 4 waves 7.21, 6-wave workgroups 8.7 (they do not spread evenly over the four SIMDs).
// fp32: 4 waves, two workgroups per CU.
// (A/B build: the unpaired bf16 form may be launched with up to SO3X_AB_UNPAIRED_THREADS threads -- 768 = three waves per SIMD
//  at <= 168 registers, 1024 = four at <= 128 -- to measure what occupancy buys a kernel bound by the vector issue port)
#ifndef SO3X_AB_UNPAIRED_THREADS
#define SO3X_AB_UNPAIRED_THREADS 512
#endif
template <int PREC, bool PAIR = true> constexpr int chain_threads() {       // launch bound
  return PREC == SO3X_PREC_BF16 ? (PAIR ? 512 : SO3X_AB_UNPAIRED_THREADS) : 256;
}
template <int PREC> constexpr int chain_threads_default() { return PREC == SO3X_PREC_BF16 ? 512 : 256; }
// WIDE: the SiLU table in its lane-replicated 64 KB form at LDS address 0, the weight image behind it (so3x_mlp.hpp).
// F16: the paired stream's f16-operand leg (SO3X_PREC_F16: the same images with IEEE half bits, so3x_mlp.hpp)
template <int PREC, bool FAST, bool PAIR, bool WIDE, bool F16 = false>
__global__ void __launch_bounds__((chain_threads<PREC, PAIR>()), (chain_threads<PREC, PAIR>() > 512 ? chain_threads<PREC, PAIR>() / 256 : 2))
k_p_sample_chain(const void* __restrict__ gimg, const float* __restrict__ beff_tab, const bf16x8* __restrict__ l0t_tab,
                 const float* __restrict__ sched, int T, const float* __restrict__ trap_p,
                 const uint16_t* __restrict__ guide_p, const float* __restrict__ x_in, float* __restrict__ x_out, int t_start,
                 const int64_t* __restrict__ t_dev, int n_steps, const float* __restrict__ axes, const float* __restrict__ unif, uint64_t seed,
                 uint64_t rng_offset, int64_t index_base, int64_t n, const char* __restrict__ cdf_rec, uint64_t* __restrict__ clk) {
  if (t_dev) {  // the first timestep read on the device (the caller's `t` tensor: no host copy, no synchronisation), clamped into the tables
    const int64_t tv = t_dev[0];
    t_start = (int)(tv < n_steps - 1 ? n_steps - 1 : (tv > T - 1 ? T - 1 : tv));
  }
  static_assert(!WIDE || (PAIR && PREC == SO3X_PREC_BF16), "the wide table belongs to the paired bf16 stream");
  static_assert(!F16 || WIDE, "the f16 operands are a leg of the shipped (paired, wide-table) stream");
  extern __shared__ __attribute__((aligned(16))) char lds_all[];
  char* lds = lds_all + (WIDE ? kWideTabBytes : 0);
  load_image(gimg, lds, image_bytes<PREC, CHAIN>());
  if constexpr (WIDE) {
    typedef __attribute__((address_space(3))) char* lds_cp;
    if ((uint32_t)(uintptr_t)(lds_cp)lds_all != 0u) __builtin_trap();  // the byte-insert addressing needs the table at LDS address 0
    fill_wide_tab(reinterpret_cast<const char*>(gimg) + (size_t)n_frags<PREC, CHAIN>() * frag_bytes<PREC>());
  }
  // WIDE: the knots of the IGSO(3) CDF (shared by every timestep) and one 4.5-KB record buffer per wave behind the image
  char* knots_lds = lds + image_bytes<PREC, CHAIN>();
  char* rowbuf = knots_lds + kChainKnotsBytes + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) * kCdfRec;
  const bool staged = WIDE && cdf_rec != nullptr && axes == nullptr;
  if (staged)
    for (int i = threadIdx.x; i < 1000; i += blockDim.x) reinterpret_cast<float*>(knots_lds)[i] = SO3X_KNOTS_DATA[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, h = lane >> 5;
  const uint32_t lt = wide_tab_lane(lane);
  const int64_t nchunks = (n + 63) / 64;
  const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
  uint64_t clk0 = 0, ref0 = 0;
  const bool stamping = clk != nullptr && wave == 0;  // wave-uniform
  if (stamping) {
    clk0 = __builtin_amdgcn_s_memtime();
    ref0 = __builtin_amdgcn_s_memrealtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0) here, so that no stamp is outstanding when the step loop counts its LDS reads
  }
  for (int64_t chunk = wave; chunk < nchunks; chunk += nwaves) {
    const int64_t idx = chunk * 64 + lane;
    const bool live = idx < n;
    const int64_t idc = live ? idx : n - 1;
    float R[9];
    load_rot9(x_in, idc, R);
    Quat q = quat_from_rmat(R);  // the state lives as a unit quaternion for the n_steps of this launch
    bf16x8 w0[3];  // this step's three layer-0 fragments, fetched a step ahead (an L2 round trip off every step's critical path)
    if constexpr (PAIR) {
#pragma unroll
      for (int k = 0; k < 3; k++) w0[k] = l0t_tab[(size_t)t_start * 192 + 64 * k + lane];
      // once per chunk: the step loop is entered with nothing the compiler tracks in flight (see below).  A use of the loaded
      // registers, not a wait instruction: the wait-count pass drops an explicit wait it deems early and re-inserts it at the use
      asm volatile("" :: "v"(w0[0]), "v"(w0[1]), "v"(w0[2]));
    }
#pragma unroll 1
    for (int s = 0; s < n_steps; s++) {
      const int t = t_start - s;
      // ---- score network: v = RotPredict(x, t)  (diffusion.py:309)
      if (s > 0) rmat_from_quat(q, R);
      const float* beff = beff_tab + (size_t)t * 96;
      float va[3], vb[3], v[3];
      StepCoef coef{};
      if constexpr (PREC == SO3X_PREC_BF16) {  // layer 0 from this timestep's fragments (bias in the K dimension)
        const bf16x8* l0t = l0t_tab + (size_t)t * 192;
        if constexpr (PAIR) {
          // the step's schedule scalars: scalar loads issued here, a network away from their use
          coef = StepCoef{sched[S_RECIP * T + t], sched[S_RECIPM1 * T + t], sched[S_COEF1 * T + t], sched[S_COEF2 * T + t]};
          if (staged) {  // this step's CDF record -> LDS (five 1-KB / 512-B DMAs); it lands while the network runs
            const char* src = cdf_rec + (size_t)t * kCdfRec + lane * 16;
            {
              typedef __attribute__((address_space(3))) char* lds_cp;
              const uint32_t dst = (uint32_t)(uintptr_t)(lds_cp)rowbuf;  // wave-uniform LDS byte address of this wave's record buffer
              // M0 = LDS base; the instruction adds its immediate offset to the global AND the LDS address, and 16 bytes per lane.
              // The fifth piece (512 B, offset beyond the 13-bit immediate) is the lower 32 lanes'.
              asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off\n\t"
                           "global_load_lds_dwordx4 %0, off offset:1024\n\t"
                           "global_load_lds_dwordx4 %0, off offset:2048\n\t"
                           "global_load_lds_dwordx4 %0, off offset:3072"
                           :: "v"(src), "s"(dst) : "memory", "m0", "scc");
              if (lane < 32)  // (the immediate offset is 13 bits signed: the fifth piece takes its own pointer)
                asm volatile("s_add_u32 m0, %1, 0x1000\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off"
                             :: "v"(src + 4096), "s"(dst) : "memory", "m0", "scc");
            }
          }
          const bf16x8* l0n = l0t_tab + (size_t)(s + 1 < n_steps ? t - 1 : t) * 192;
          forward_pair_bf16<WIDE, F16>(lds, R, w0, va, vb, lane, lt, l0n);  // both tiles as one software-pipelined stream; reloads w0 for the next step
        } else {
          forward_tile<PREC, CHAIN, 1, true>(lds, R, nullptr, 0, nullptr, va, lane, l0t);
          forward_tile<PREC, CHAIN, 2, true>(lds, R, nullptr, 0, nullptr, vb, lane, l0t);
        }
      } else {
        forward_tile<PREC, CHAIN, 1>(lds, R, beff, 0, nullptr, va, lane);  // tile A = samples 0..31 of the chunk
        forward_tile<PREC, CHAIN, 2>(lds, R, beff, 0, nullptr, vb, lane);  // tile B = samples 32..63
      }
#pragma unroll
      for (int j = 0; j < 3; j++) {  // tile B results sit in lanes 0..31, their owners are lanes 32..63
        if constexpr (PAIR) {  // one v_permlane32_swap: [va of the lower half | vb of the lower half] (no LDS exchange, no select)
          v[j] = __builtin_bit_cast(float, __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(uint32_t, va[j]), __builtin_bit_cast(uint32_t, vb[j]), false, false)[0]);
          continue;
        }
        const float o = __shfl_xor(vb[j], 32);
        v[j] = h ? o : va[j];
      }
      // ---- posterior mean + noise (diffusion.py:291-326), so3x_reverse_step.hpp
      // the record has landed (and is visible to this wave's LDS reads).  As the BUILTIN, which the compiler's wait-count pass
      // reads, and on EVERY path to the loop's back edge: it then knows that nothing it tracks -- the next step's layer-0
      // fragments, requested behind this step's layer 0 -- is still in flight, and puts no wait in front of the next step's first
      // MFMAs, where one would also cover the record DMA issued just before them (0x0F70 = vmcnt(0) alone)
      __builtin_amdgcn_s_waitcnt(0x0F70);
      asm volatile("" ::: "memory");
      if (staged) {
        q = reverse_step<FAST>(q, v, sched, T, t, trap_p, guide_p, axes, unif, idc, seed, rng_offset, (uint64_t)(index_base + idx),
                               reinterpret_cast<const float*>(rowbuf),
                               guide_p ? reinterpret_cast<const uint16_t*>(rowbuf + kCdfGuideOff) : nullptr,
                               reinterpret_cast<const float*>(knots_lds), &coef);
      } else {
        q = reverse_step<FAST>(q, v, sched, T, t, trap_p, guide_p, axes, unif, idc, seed, rng_offset, (uint64_t)(index_base + idx));
      }
    }
    rmat_from_quat(qnormalize(q), R);
    if (live) store_rot9(x_out, idx, R);
  }
  if (stamping) {
    const uint64_t clk1 = __builtin_amdgcn_s_memtime(), ref1 = __builtin_amdgcn_s_memrealtime();
    if (lane == 0) { clk[0] = clk1 - clk0; clk[1] = ref1 - ref0; }
  }
}

// The product launcher takes no switch from anywhere but its arguments: ONE bf16 form (hardware sine / cosine, paired stream,
// lane-replicated table, LDS-staged CDF record, 8-wave workgroups) and the fp32 parity form.  -DSO3X_AB_BUILD (the second
// library libso3x_ab.so that tools/ab and tests/test_gpu_parity.py::test_chain_bf16_kernel_forms_agree load explicitly; never the
// library the package loads) adds the other forms behind environment switches read per launch: SO3X_AB_TRIG=cw forces the
// Cody-Waite sine / cosine in the bf16 kernel, SO3X_AB_PAIR=0 one tile after the other, SO3X_AB_TAB=narrow the 2 KB table,
// SO3X_AB_CDF=global the search on global memory, SO3X_AB_BLOCK=<threads> the workgroup size (multiple of 64 up to 512).
#ifdef SO3X_AB_BUILD
inline int ab_env(const char* name, const char* value) {
  const char* e = getenv(name);
  return e && !strcmp(e, value);
}
#endif

template <int PREC, bool FAST, bool PAIR, bool WIDE = false, bool F16 = false>
int launch_chain_v(hipStream_t s, const void* ws, const float* beff, const float* sched, int T, const float* trap_p,
                   const uint16_t* guide_p, const float* x_in, float* x_out, int t_start, const int64_t* t_dev, int n_steps, const float* axes,
                   const float* unif, uint64_t seed, uint64_t rng_offset, int64_t index_base, int64_t n) {
  constexpr int IMG = image_bytes<PREC, CHAIN>() + (WIDE ? kWideTabBytes + kChainKnotsBytes + kChainRowBufBytes : 0);
  int max_blocks = 0;
  int threads = chain_threads_default<PREC>();
  bool staged_cdf = WIDE;
#ifdef SO3X_AB_BUILD
  if (PREC == SO3X_PREC_BF16 && getenv("SO3X_AB_BLOCK")) threads = atoi(getenv("SO3X_AB_BLOCK"));
  if (threads < 64 || threads > chain_threads<PREC, PAIR>() || threads % 64) return SO3X_ERR_INVALID_ARG;
  if (ab_env("SO3X_AB_CDF", "global")) staged_cdf = false;
  static PerDevice residents[17];  // one cache per workgroup size
  PerDevice& resident = residents[threads / 64];
#else
  static PerDevice resident;
#endif
  if (int rc = resident_blocks(resident, reinterpret_cast<const void*>(&k_p_sample_chain<PREC, FAST, PAIR, WIDE, F16>), threads, IMG, &max_blocks)) return rc;
  const int64_t nchunks = (n + 63) / 64;
  const int wpb = threads / 64;
  const int64_t want = (nchunks + wpb - 1) / wpb;
  const int grid = (int)(want < max_blocks ? want : max_blocks);
  const bf16x8* l0t = PREC == SO3X_PREC_BF16 ? reinterpret_cast<const bf16x8*>(reinterpret_cast<const char*>(ws) + l0t_offset(T)) : nullptr;
  hipLaunchKernelGGL((k_p_sample_chain<PREC, FAST, PAIR, WIDE, F16>), dim3(grid), dim3(threads), IMG, s, ws, beff, l0t, sched, T, trap_p, guide_p, x_in, x_out,
                     t_start, t_dev, n_steps, axes, unif, seed, rng_offset, index_base, n,
                     staged_cdf ? reinterpret_cast<const char*>(ws) + cdf_offset(T) : nullptr,
                     reinterpret_cast<uint64_t*>(const_cast<char*>(reinterpret_cast<const char*>(ws)) +
                                                 (PREC == SO3X_PREC_BF16 ? clock_offset_bf16(T) : clock_offset_f32(T))));
  return check_launch();
}

template <int PREC>
int launch_chain(hipStream_t s, const void* ws, const float* beff, const float* sched, int T, const float* trap_p,
                 const uint16_t* guide_p, const float* x_in, float* x_out, int t_start, const int64_t* t_dev, int n_steps, const float* axes,
                 const float* unif, uint64_t seed, uint64_t rng_offset, int64_t index_base, int64_t n, bool f16 = false) {
#define SO3X_CHAIN_ARGS s, ws, beff, sched, T, trap_p, guide_p, x_in, x_out, t_start, t_dev, n_steps, axes, unif, seed, rng_offset, index_base, n
  if constexpr (PREC == SO3X_PREC_BF16) {
    if (f16) return launch_chain_v<PREC, true, true, true, true>(SO3X_CHAIN_ARGS);
#ifdef SO3X_AB_BUILD
    if (ab_env("SO3X_AB_TRIG", "cw")) return launch_chain_v<PREC, false, true, true>(SO3X_CHAIN_ARGS);
    if (ab_env("SO3X_AB_PAIR", "0")) return launch_chain_v<PREC, true, false>(SO3X_CHAIN_ARGS);
    if (ab_env("SO3X_AB_TAB", "narrow")) return launch_chain_v<PREC, true, true>(SO3X_CHAIN_ARGS);
#endif
    return launch_chain_v<PREC, true, true, true>(SO3X_CHAIN_ARGS);
  } else {
    return launch_chain_v<PREC, false, false>(SO3X_CHAIN_ARGS);
  }
#undef SO3X_CHAIN_ARGS
}

}  // namespace

namespace so3x {
// so3x_q_sample_target with the option of drawing the timesteps in the kernel (t == nullptr, t_draw = where they go)
int launch_q_sample_target(hipStream_t s, const float* sched, int T, const float* trap_q, const uint16_t* guide_q, const float* x0,
                           const int64_t* t, int64_t* t_draw, int quirk_col0, const float* noise_in, const float* axes,
                           const float* unif, uint64_t seed, uint64_t rng_offset, const int64_t* rng_offset_dev, int64_t index_base,
                           float* x_t, float* target, float* noise_out, int64_t n) {
  if (n < 0 || T <= 0 || (n && (!sched || !x0 || (!t && !t_draw))) || (n && !noise_in && !trap_q) ||
      ((axes == nullptr) != (unif == nullptr)))
    return SO3X_ERR_INVALID_ARG;
  if (n == 0) return SO3X_OK;
  const int64_t nt64 = (n + kWave - 1) / kWave;
  int64_t want = (nt64 + 3) / 4;   // one tile per wave
  if (want > (1 << 20)) want = 1 << 20;
  if (!noise_in && !axes)
    hipLaunchKernelGGL(k_q_sample_target<true>, dim3((unsigned)want), dim3(kBlock), 0, s, sched, T, trap_q, guide_q, x0, t, t_draw, quirk_col0,
                       noise_in, axes, unif, seed, rng_offset, rng_offset_dev, index_base, x_t, target, noise_out, n);
  else
    hipLaunchKernelGGL(k_q_sample_target<false>, dim3((unsigned)want), dim3(kBlock), 0, s, sched, T, trap_q, guide_q, x0, t, t_draw, quirk_col0,
                       noise_in, axes, unif, seed, rng_offset, rng_offset_dev, index_base, x_t, target, noise_out, n);
  return check_launch();
}
}  // namespace so3x

extern "C" {

int so3x_q_sample_target(so3x_stream_t s, const float* sched, int T, const float* trap_q, const uint16_t* guide_q,
                         const float* x0, const int64_t* t,
                         int quirk_col0, const float* noise_in, const float* axes, const float* unif, uint64_t seed,
                         uint64_t rng_offset, const int64_t* rng_offset_dev, int64_t index_base, float* x_t, float* target,
                         float* noise_out, int64_t n) {
  if (n && !t) return SO3X_ERR_INVALID_ARG;
  return so3x::launch_q_sample_target((hipStream_t)s, sched, T, trap_q, guide_q, x0, t, nullptr, quirk_col0, noise_in, axes, unif, seed,
                                      rng_offset, rng_offset_dev, index_base, x_t, target, noise_out, n);
}

int so3x_p_mean(so3x_stream_t s, const float* sched, int T, const float* x, const float* v, int t, float* x0hat,
                float* mean, int64_t n) {
  if (n < 0 || T <= 0 || t < 0 || t >= T || (n && (!sched || !x || !v || !mean))) return SO3X_ERR_INVALID_ARG;
  if (n == 0) return SO3X_OK;
  hipLaunchKernelGGL(k_p_mean, dim3(grid_for_tiles((n + kTile - 1) / kTile)), dim3(kBlock), 0, (hipStream_t)s, sched, T, x, v,
                     t, (const int64_t*)nullptr, (int64_t)0, x0hat, mean, n);
  return check_launch();
}

int so3x_p_mean_t(so3x_stream_t s, const float* sched, int T, const float* x, const float* v, const int64_t* t, int64_t t_stride,
                  float* x0hat, float* mean, int64_t n) {
  if (n < 0 || T <= 0 || (t_stride != 0 && t_stride != 1) || (n && (!sched || !x || !v || !mean || !t))) return SO3X_ERR_INVALID_ARG;
  if (n == 0) return SO3X_OK;
  hipLaunchKernelGGL(k_p_mean, dim3(grid_for_tiles((n + kTile - 1) / kTile)), dim3(kBlock), 0, (hipStream_t)s, sched, T, x, v,
                     0, t, t_stride, x0hat, mean, n);
  return check_launch();
}

size_t so3x_p_sample_workspace_bytes(int T, int precision) {
  const int p = precision == SO3X_PREC_F32 ? SO3X_PREC_F32 : SO3X_PREC_BF16;
  const int Tn = T > 0 ? T : 0;
  return (p == SO3X_PREC_BF16 ? clock_offset_bf16(Tn) : clock_offset_f32(Tn)) + kClockBytes;
}

size_t so3x_p_sample_clock_offset(int T, int precision) {
  const int Tn = T > 0 ? T : 0;
  return precision == SO3X_PREC_F32 ? clock_offset_f32(Tn) : clock_offset_bf16(Tn);
}

// the launch-independent state of the chain kernel for ALL T timesteps: weight image, per-timestep effective-bias rows and layer-0
// fragments, CDF records
static int prepare_steps(hipStream_t s, const float* params, int T, const float* trap_p, const uint16_t* guide_p, int precision, bool f16,
                         void* workspace, int t_first, int n_steps) {
  int rc = launch_prep(s, params, precision, CHAIN, T, workspace, 3, nullptr, true, nullptr, t_first, n_steps, f16);
  if (rc) return rc;
  if (precision == SO3X_PREC_BF16 && (rc = launch_prep_l0t(s, params, T, workspace, t_first, n_steps, f16))) return rc;
  if (precision == SO3X_PREC_BF16) {  // the CDF records of these steps (LDS-DMA source of the chain kernel)
    hipLaunchKernelGGL(k_prep_cdf, dim3(n_steps), dim3(256), 0, s, trap_p, guide_p, reinterpret_cast<char*>(workspace) + cdf_offset(T), t_first);
    if ((rc = check_launch())) return rc;
  }
  return SO3X_OK;
}
static int run_steps(hipStream_t s, const float* sched, int T, const float* trap_p, const uint16_t* guide_p, const float* x_in, float* x_out,
                     int t_start, const int64_t* t_dev, int n_steps, const float* axes, const float* unif, uint64_t seed, uint64_t rng_offset,
                     int64_t index_base, int64_t n, int precision, bool f16, void* workspace) {
  const float* beff = reinterpret_cast<const float*>(reinterpret_cast<const char*>(workspace) + beff_offset(precision, CHAIN));
  if (precision == SO3X_PREC_F32)
    return launch_chain<SO3X_PREC_F32>(s, workspace, beff, sched, T, trap_p, guide_p, x_in, x_out, t_start, t_dev, n_steps, axes, unif, seed,
                                       rng_offset, index_base, n);
  return launch_chain<SO3X_PREC_BF16>(s, workspace, beff, sched, T, trap_p, guide_p, x_in, x_out, t_start, t_dev, n_steps, axes, unif, seed,
                                      rng_offset, index_base, n, f16);
}

int so3x_p_sample_chain(so3x_stream_t s, const float* params, const float* sched, int T, const float* trap_p,
                        const uint16_t* guide_p, const float* x_in, float* x_out, int t_start, int n_steps, const float* axes, const float* unif,
                        uint64_t seed, uint64_t rng_offset, int64_t index_base, int64_t n, int precision, void* workspace,
                        size_t workspace_bytes) {
  if (n < 0 || T <= 0 || n_steps < 0 || t_start < 0 || t_start >= T || t_start - n_steps + 1 < 0 ||
      (n && (!params || !sched || !trap_p || !x_in || !x_out)) || ((axes == nullptr) != (unif == nullptr)) ||
      (axes && n_steps > 1))
    return SO3X_ERR_INVALID_ARG;
  if (precision != SO3X_PREC_F32 && precision != SO3X_PREC_BF16 && precision != SO3X_PREC_F16) return SO3X_ERR_UNSUPPORTED;
  if (!workspace || workspace_bytes < so3x_p_sample_workspace_bytes(T, precision)) return SO3X_ERR_WORKSPACE;
  if (n == 0 || n_steps == 0) return SO3X_OK;
  const bool f16 = precision == SO3X_PREC_F16;   // the bf16 path's images, tables and kernel with IEEE half operand bits
  if (f16) precision = SO3X_PREC_BF16;
  // image + the per-timestep rows of the steps this launch runs (t_start - n_steps + 1 .. t_start)
  const int t_first = t_start - n_steps + 1;
  if (int rc = prepare_steps((hipStream_t)s, params, T, trap_p, guide_p, precision, f16, workspace, t_first, n_steps)) return rc;
  return run_steps((hipStream_t)s, sched, T, trap_p, guide_p, x_in, x_out, t_start, nullptr, n_steps, axes, unif, seed, rng_offset, index_base, n,
                   precision, f16, workspace);
}

int so3x_p_sample_prepare(so3x_stream_t s, const float* params, int T, const float* trap_p, const uint16_t* guide_p, int precision,
                          void* workspace, size_t workspace_bytes) {
  if (T <= 0 || !params || !trap_p) return SO3X_ERR_INVALID_ARG;
  if (precision != SO3X_PREC_F32 && precision != SO3X_PREC_BF16 && precision != SO3X_PREC_F16) return SO3X_ERR_UNSUPPORTED;
  if (!workspace || workspace_bytes < so3x_p_sample_workspace_bytes(T, precision)) return SO3X_ERR_WORKSPACE;
  const bool f16 = precision == SO3X_PREC_F16;
  return prepare_steps((hipStream_t)s, params, T, trap_p, guide_p, f16 ? SO3X_PREC_BF16 : precision, f16, workspace, 0, T);
}

int so3x_p_sample_prepared(so3x_stream_t s, const float* sched, int T, const float* trap_p, const uint16_t* guide_p, const float* x_in,
                           float* x_out, int t_start, const int64_t* t_dev, int n_steps, const float* axes, const float* unif, uint64_t seed,
                           uint64_t rng_offset, int64_t index_base, int64_t n, int precision, void* workspace, size_t workspace_bytes) {
  if (n < 0 || T <= 0 || n_steps < 0 || n_steps > T || (n && (!sched || !trap_p || !x_in || !x_out)) || ((axes == nullptr) != (unif == nullptr)) ||
      (axes && n_steps > 1) || (!t_dev && (t_start < 0 || t_start >= T || t_start - n_steps + 1 < 0)))
    return SO3X_ERR_INVALID_ARG;
  if (precision != SO3X_PREC_F32 && precision != SO3X_PREC_BF16 && precision != SO3X_PREC_F16) return SO3X_ERR_UNSUPPORTED;
  if (!workspace || workspace_bytes < so3x_p_sample_workspace_bytes(T, precision)) return SO3X_ERR_WORKSPACE;
  if (n == 0 || n_steps == 0) return SO3X_OK;
  const bool f16 = precision == SO3X_PREC_F16;
  return run_steps((hipStream_t)s, sched, T, trap_p, guide_p, x_in, x_out, t_dev ? 0 : t_start, t_dev, n_steps, axes, unif, seed, rng_offset,
                   index_base, n, f16 ? SO3X_PREC_BF16 : precision, f16, workspace);
}

}  // extern "C"
