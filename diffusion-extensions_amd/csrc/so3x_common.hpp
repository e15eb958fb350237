// so3x_common.hpp -- launch plumbing and AoS<->register staging shared by the kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/so3x.h"

namespace so3x {

constexpr int kBlock = 256;  // 4 waves of 64
constexpr int kTile = 256;   // samples per block-iteration in the pointwise kernels

// ~8 blocks per CU on 256 CUs, grid-stride over tiles beyond that (guide: G11)
inline int grid_for_tiles(int64_t ntiles) { return (int)(ntiles < 2048 ? (ntiles < 1 ? 1 : ntiles) : 2048); }

inline int check_launch() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? SO3X_OK : (int)e;
}

// Launch facts that have to be established once PER DEVICE (hipFuncSetAttribute and occupancy are per device, and one
// process may drive several GPUs): a static table at each use site, indexed by the current device's ordinal.
// Host-only HIP calls (no sync, legal outside and -- after the first call -- never made during stream capture).
struct PerDevice { int v[64]; };
inline int per_device_slot(PerDevice& st, int** slot) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return (int)e;
  *slot = &st.v[dev & 63];
  return SO3X_OK;
}
// raise the dynamic-LDS cap of `kernel` on the current device (once)
inline int ensure_dyn_lds(PerDevice& st, const void* kernel, int bytes) {
  int* slot;
  int rc = per_device_slot(st, &slot);
  if (rc) return rc;
  if (__atomic_load_n(slot, __ATOMIC_ACQUIRE)) return SO3X_OK;
  hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e != hipSuccess) return (int)e;
  __atomic_store_n(slot, 1, __ATOMIC_RELEASE);
  return SO3X_OK;
}
// ... and the number of workgroups of `kernel` the current device keeps resident (occupancy x CUs), for persistent grids
inline int resident_blocks(PerDevice& st, const void* kernel, int threads, int lds_bytes, int* cap) {
  int* slot;
  int rc = per_device_slot(st, &slot);
  if (rc) return rc;
  int c = __atomic_load_n(slot, __ATOMIC_ACQUIRE);
  if (!c) {
    hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    if (e != hipSuccess) return (int)e;
    int per_cu = 0, dev = 0, cus = 0;
    if ((e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, threads, lds_bytes)) != hipSuccess) return (int)e;
    if ((e = hipGetDevice(&dev)) != hipSuccess) return (int)e;
    if ((e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev)) != hipSuccess) return (int)e;
    c = (per_cu > 0 ? per_cu : 1) * (cus > 0 ? cus : 256);
    __atomic_store_n(slot, c, __ATOMIC_RELEASE);
  }
  *cap = c;
  return SO3X_OK;
}

// The reference's tensors are AoS [n][W] fp32 (W = 9 rotations, 3 vectors, 4 quats):
// a lane-per-sample access would be a 36-B-strided gather.  Instead the block moves
// the tile's W*256 contiguous floats with 16-B-per-lane coalesced loads into LDS and
// each lane then reads its W floats at stride W (W odd -> bank-conflict free).
template <int W>
__device__ __forceinline__ void tile_to_lds(const float* __restrict__ g, int64_t base, int cnt, float* lds) {
  const float* src = g + base * W;
  if (cnt == kTile && ((reinterpret_cast<uintptr_t>(src) & 15) == 0)) {
    const float4* s4 = reinterpret_cast<const float4*>(src);
    float4* d4 = reinterpret_cast<float4*>(lds);
#pragma unroll
    for (int i = threadIdx.x; i < W * (kTile / 4); i += kBlock) d4[i] = s4[i];
  } else {
    for (int i = threadIdx.x; i < cnt * W; i += kBlock) lds[i] = src[i];
  }
}

template <int W>
__device__ __forceinline__ void lds_to_tile(float* __restrict__ g, int64_t base, int cnt, const float* lds) {
  float* dst = g + base * W;
  if (cnt == kTile && ((reinterpret_cast<uintptr_t>(dst) & 15) == 0)) {
    float4* d4 = reinterpret_cast<float4*>(dst);
    const float4* s4 = reinterpret_cast<const float4*>(lds);
#pragma unroll
    for (int i = threadIdx.x; i < W * (kTile / 4); i += kBlock) d4[i] = s4[i];
  } else {
    for (int i = threadIdx.x; i < cnt * W; i += kBlock) dst[i] = lds[i];
  }
}

// Load this lane's W floats of tile sample threadIdx.x (block-wide; contains barriers).
template <int W>
__device__ __forceinline__ void load_rows(const float* __restrict__ g, int64_t base, int cnt, float* lds, float* r) {
  __syncthreads();  // previous users of lds are done
  tile_to_lds<W>(g, base, cnt, lds);
  __syncthreads();
#pragma unroll
  for (int j = 0; j < W; j++) r[j] = lds[threadIdx.x * W + j];
}

template <int W>
__device__ __forceinline__ void store_rows(float* __restrict__ g, int64_t base, int cnt, float* lds, const float* r) {
  __syncthreads();
#pragma unroll
  for (int j = 0; j < W; j++) lds[threadIdx.x * W + j] = r[j];
  __syncthreads();
  lds_to_tile<W>(g, base, cnt, lds);
}

// ---- wave-private variant: one wave = 64 samples, no workgroup barrier -----------------
// The same coalesced AoS<->lane transposition, but through a per-wave LDS slice (W*64
// floats).  A wave's LDS operations complete in issue order, so only the compiler has to
// be told not to reorder them (wave_barrier); waves of a workgroup never wait for each other.
constexpr int kWave = 64;

template <int W>
__device__ __forceinline__ void wave_load_rows(const float* __restrict__ g, int64_t base, int cnt, float* wlds, float* r) {
  const int lane = threadIdx.x & 63;
  const float* src = g + base * W;
  __builtin_amdgcn_wave_barrier();
  if (cnt == kWave && ((reinterpret_cast<uintptr_t>(src) & 15) == 0)) {
    const float4* s4 = reinterpret_cast<const float4*>(src);
    float4* d4 = reinterpret_cast<float4*>(wlds);
#pragma unroll
    for (int i = 0; i < (W * 16 + 63) / 64; i++) {
      const int k = lane + 64 * i;
      if (k < W * 16) d4[k] = s4[k];
    }
  } else {
    for (int i = lane; i < cnt * W; i += 64) wlds[i] = src[i];
  }
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int j = 0; j < W; j++) r[j] = wlds[lane * W + j];
  __builtin_amdgcn_wave_barrier();
}

// NT: non-temporal stores for outputs that are written once and not read back by this kernel (streaming kernels)
typedef float f32x4_nt __attribute__((ext_vector_type(4)));
template <int W, bool NT = false>
__device__ __forceinline__ void wave_store_rows(float* __restrict__ g, int64_t base, int cnt, float* wlds, const float* r) {
  const int lane = threadIdx.x & 63;
  float* dst = g + base * W;
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int j = 0; j < W; j++) wlds[lane * W + j] = r[j];
  __builtin_amdgcn_wave_barrier();
  if (cnt == kWave && ((reinterpret_cast<uintptr_t>(dst) & 15) == 0)) {
    float4* d4 = reinterpret_cast<float4*>(dst);
    const float4* s4 = reinterpret_cast<const float4*>(wlds);
#pragma unroll
    for (int i = 0; i < (W * 16 + 63) / 64; i++) {
      const int k = lane + 64 * i;
      if (k < W * 16) {
        if constexpr (NT) __builtin_nontemporal_store(reinterpret_cast<const f32x4_nt*>(wlds)[k], reinterpret_cast<f32x4_nt*>(dst) + k);
        else d4[k] = s4[k];
      }
    }
  } else {
    for (int i = lane; i < cnt * W; i += 64) dst[i] = wlds[i];
  }
  __builtin_amdgcn_wave_barrier();
}

// Split form of wave_load_rows<9> for software prefetch: issue the NEXT tile's global loads into
// registers (9 VGPRs) before computing the current tile, land them in LDS when their turn comes.
// Only for full, 16-B-aligned tiles (the caller falls back to wave_load_rows otherwise).
struct Pref9 { float4 a, b, c; };
template <bool NT = false>
__device__ __forceinline__ Pref9 wave_prefetch9(const float* __restrict__ g, int64_t base) {
  const int lane = threadIdx.x & 63;
  const float4* s4 = reinterpret_cast<const float4*>(g + base * 9);
  Pref9 p;
  if constexpr (NT) {  // read-once input of a streaming kernel
    const f32x4_nt* v4 = reinterpret_cast<const f32x4_nt*>(g + base * 9);
    const f32x4_nt a = __builtin_nontemporal_load(v4 + lane), b = __builtin_nontemporal_load(v4 + lane + 64);
    p.a = float4{a[0], a[1], a[2], a[3]};
    p.b = float4{b[0], b[1], b[2], b[3]};
    if (lane < 16) { const f32x4_nt c = __builtin_nontemporal_load(v4 + lane + 128); p.c = float4{c[0], c[1], c[2], c[3]}; }
    else p.c = float4{0.f, 0.f, 0.f, 0.f};
  } else {
    p.a = s4[lane];
    p.b = s4[lane + 64];
    p.c = lane < 16 ? s4[lane + 128] : float4{0.f, 0.f, 0.f, 0.f};
  }
  return p;
}
__device__ __forceinline__ void wave_commit9(const Pref9& p, float* wlds, float* r) {
  const int lane = threadIdx.x & 63;
  float4* d4 = reinterpret_cast<float4*>(wlds);
  __builtin_amdgcn_wave_barrier();
  d4[lane] = p.a;
  d4[lane + 64] = p.b;
  if (lane < 16) d4[lane + 128] = p.c;
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int j = 0; j < 9; j++) r[j] = wlds[lane * 9 + j];
  __builtin_amdgcn_wave_barrier();
}

// The same prefetch with NO registers held: the tile's 2,304 bytes go global -> LDS by LDS-DMA (three global_load_lds_dwordx4:
// 64 + 64 + 16 lanes x 16 B), asynchronously; wave_dma9_commit waits for them (vmcnt) and reads the lane's nine floats.
// Full, 16-byte-aligned tiles only.  The staging buffer must not be touched in between.
__device__ __forceinline__ void wave_dma9(const float* __restrict__ g, int64_t base, float* wlds) {
  const int lane = threadIdx.x & 63;
  const char* src = reinterpret_cast<const char*>(g + base * 9) + lane * 16;
  char* dst = reinterpret_cast<char*>(wlds);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this wave's earlier reads of the buffer are done before new data may land
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + 1024),
                                   (__attribute__((address_space(3))) void*)(dst + 1024), 16, 0, 0);
  if (lane < 16)
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + 2048),
                                     (__attribute__((address_space(3))) void*)(dst + 2048), 16, 0, 0);
}
__device__ __forceinline__ void wave_dma9_commit(const float* wlds, float* r) {
  const int lane = threadIdx.x & 63;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the DMA pieces have landed and are visible to this wave's LDS reads
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int j = 0; j < 9; j++) r[j] = wlds[lane * 9 + j];
  __builtin_amdgcn_wave_barrier();
}

// A lane's whole rotation (36 contiguous bytes, 4-byte aligned) moved with three wide accesses
// (dwordx4 + dwordx3 + dwordx2) instead of nine 4-byte ones at a 36-byte lane stride: the nine
// partial-line stores showed up as 5.8x WRITE_SIZE inflation in the PMC pass of round 1.
struct __attribute__((packed, aligned(4))) Rot9 { float m[9]; };
__device__ __forceinline__ void load_rot9(const float* __restrict__ g, int64_t idx, float* r) {
  const Rot9 v = *reinterpret_cast<const Rot9*>(g + idx * 9);
#pragma unroll
  for (int j = 0; j < 9; j++) r[j] = v.m[j];
}
__device__ __forceinline__ void store_rot9(float* __restrict__ g, int64_t idx, const float* r) {
  Rot9 v;
#pragma unroll
  for (int j = 0; j < 9; j++) v.m[j] = r[j];
  *reinterpret_cast<Rot9*>(g + idx * 9) = v;
}

// scalar-per-sample operand with stride 0 (broadcast) or 1
__device__ __forceinline__ float load_scalar(const float* p, int64_t stride, int64_t i) { return p[i * stride]; }

// so3x_diffusion.hip: so3x_q_sample_target with the option of drawing the timesteps in the kernel (used by so3x_train_fwd)
int launch_q_sample_target(hipStream_t s, const float* sched, int T, const float* trap_q, const uint16_t* guide_q, const float* x0,
                           const int64_t* t, int64_t* t_draw, int quirk_col0, const float* noise_in, const float* axes,
                           const float* unif, uint64_t seed, uint64_t rng_offset, const int64_t* rng_offset_dev, int64_t index_base,
                           float* x_t, float* target, float* noise_out, int64_t n);

}  // namespace so3x
