// LDS store / transposed-read throughput on gfx950 for the address patterns of the one-kernel training step's hand-over images
// (tools/ab: measurement only).   hipcc -O3 --offload-arch=gfx950 tools/ab/lds_bench.hip -o build/lds_bench && build/lds_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

constexpr int REPS = 64, NI = 16;
// pattern p: byte offset of lane's access number i (i < NI) within a 48 KB window
__device__ __forceinline__ int addr_of(int p, int lane, int i) {
  const int r = lane & 31, h = lane >> 5;
  switch (p) {
    case 0: return lane * 8 + i * 512;                                                        // linear b64
    case 1: { const int sw = (r & 7) | ((((r >> 1) ^ (r >> 3)) & 1) << 3); return r * 384 + (((2 * i + h) * 8) ^ (8 * sw)); }   // shipped
    case 2: { const int sw = ((r >> 2) & 7) | (((r >> 1) & 1) << 3); return r * 384 + (((2 * i + h) * 8) ^ (8 * sw)); }         // bijective in r>>1
    case 3: { const int sw = (r & 7) | ((((r >> 1) ^ (r >> 3)) & 1) << 3); return r * 392 + (((2 * i + h) * 8) ^ (8 * sw)); }   // shipped + pitch 392
    case 4: return lane * 16 + i * 1024;                                                       // linear b128
    case 5: { const int sw = (r >> 1) & 7; return r * 384 + ((((i & 7) * 2 + h) * 16) ^ (16 * sw)) % 384; }                      // b128 rows
    default: return 0;
  }
}
template <int W>
__global__ void __launch_bounds__(512) k_store(int p, uint64_t* out, int waves_active) {
  extern __shared__ char lds[];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  int a[NI];
  for (int i = 0; i < NI; i++) a[i] = addr_of(p, lane, i) + (wid & 3) * 12544;
  __syncthreads();
  if (!((waves_active >> wid) & 1)) return;
  const uint64_t t0 = __builtin_amdgcn_s_memtime();
  for (int rep = 0; rep < REPS; rep++) {
#pragma unroll
    for (int i = 0; i < NI; i++) {
      typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
      typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
      if (W == 8) asm volatile("ds_write_b64 %0, %1" :: "v"(a[i]), "v"(u32x2{(unsigned)rep, (unsigned)i}) : "memory");
      else asm volatile("ds_write_b128 %0, %1" :: "v"(a[i]), "v"(u32x4{(unsigned)rep, (unsigned)i, 0u, 1u}) : "memory");
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const uint64_t t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) out[blockIdx.x * 8 + wid] = t1 - t0;
}
// transposed reads: pattern q = natural fragment-order weight image (4-way expected) / permuted slots / the fimg layout
__device__ __forceinline__ int raddr_of(int q, int lane, int i) {
  const int h = lane >> 5, l32 = lane & 31, G = l32 >> 4, qq = (l32 & 15) >> 2, pp = l32 & 3;
  const int part = i & 1, ks = (i >> 1) % 5, to = (i >> 1) / 5 % 2;
  const int row = 16 * (ks & 1) + 8 * part + 4 * h + qq, tout = ks >> 1, ksf = 2 * to + G, hf = pp & 1;
  int slot = row + 32 * hf;
  if (q == 1) slot ^= (4 * hf) ^ (8 * (ksf & 1));
  if (q == 2) {  // the hand-over image as shipped
    const int r = 8 * h + qq + 4 * part, sw = (r & 7) | ((((qq >> 1) ^ h) & 1) << 3);
    return r * 384 + 8 * ((4 * G + pp) ^ sw) + (i >> 1) * 128 % 256;
  }
  return ((tout * 5 + ksf) * 64 + slot) * 16 + 8 * (pp >> 1);
}
__global__ void __launch_bounds__(512) k_tr(int q, uint64_t* out, int waves_active, int* sink) {
  extern __shared__ char lds[];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 40000 / 4; i += 512) reinterpret_cast<int*>(lds)[i] = i;
  int a[NI];
  for (int i = 0; i < NI; i++) a[i] = raddr_of(q, lane, i);
  __syncthreads();
  if (!((waves_active >> wid) & 1)) return;
  typedef short s16x4 __attribute__((ext_vector_type(4)));
  typedef __attribute__((address_space(3))) s16x4* lds_p;
  int acc = 0;
  const uint64_t t0 = __builtin_amdgcn_s_memtime();
  for (int rep = 0; rep < REPS; rep++) {
    s16x4 v[NI];
#pragma unroll
    for (int i = 0; i < NI; i++) v[i] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(lds + a[i]));
#pragma unroll
    for (int i = 0; i < NI; i++) acc += v[i][0] + v[i][3];
  }
  const uint64_t t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) out[blockIdx.x * 8 + wid] = t1 - t0;
  if (acc == 12345678) *sink = acc;
}
template <int W>
__global__ void __launch_bounds__(512) k_read(uint64_t* out, int waves_active, int* sink, int gather) {
  extern __shared__ char lds[];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 40000 / 4; i += 512) reinterpret_cast<int*>(lds)[i] = i;
  int a[NI];
  for (int i = 0; i < NI; i++) a[i] = gather ? (((lane * 2654435761u + i * 40503u) >> 7) & 255) * W : lane * W + i * 64 * W;
  __syncthreads();
  if (!((waves_active >> wid) & 1)) return;
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  unsigned acc = 0;
  const uint64_t t0 = __builtin_amdgcn_s_memtime();
  for (int rep = 0; rep < REPS; rep++) {
    if (W == 4) { unsigned v[NI];
#pragma unroll
      for (int i = 0; i < NI; i++) asm volatile("ds_read_b32 %0, %1" : "=v"(v[i]) : "v"(a[i]) : "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int i = 0; i < NI; i++) acc += v[i];
    } else if (W == 8) { u32x2 v[NI];
#pragma unroll
      for (int i = 0; i < NI; i++) asm volatile("ds_read_b64 %0, %1" : "=v"(v[i]) : "v"(a[i]) : "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int i = 0; i < NI; i++) acc += v[i][0] + v[i][1];
    } else { u32x4 v[NI];
#pragma unroll
      for (int i = 0; i < NI; i++) asm volatile("ds_read_b128 %0, %1" : "=v"(v[i]) : "v"(a[i]) : "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int i = 0; i < NI; i++) acc += v[i][0] + v[i][3];
    }
  }
  const uint64_t t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) out[blockIdx.x * 8 + wid] = t1 - t0;
  if (acc == 12345678u) *sink = acc;
}
int main() {
  uint64_t* d; int* sink;
  hipMalloc(&d, 8 * sizeof(uint64_t)); hipMalloc(&sink, 4);
  uint64_t h[8];
  auto report = [&](const char* name, int bytes_per_instr, int wa) {
    hipDeviceSynchronize(); hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    double worst = 0; int cnt = 0; for (int w = 0; w < 8; w++) if ((wa >> w) & 1) { cnt++; worst = h[w] > worst ? (double)h[w] : worst; }
    // s_memtime counts at 100 MHz on this part; convert with the shader clock the caller knows -- print both raw ticks and per-instruction
    printf("%-44s wave mask 0x%02x  ticks %8.0f  ticks/instr/wave %.3f  aggregate B/tick %.1f\n", name, wa, worst, worst / (REPS * NI),
           (double)cnt * REPS * NI * bytes_per_instr / worst);
  };
  const char* sn[] = {"store b64 linear", "store b64 image (shipped swizzle)", "store b64 image (swizzle r>>1)", "store b64 image (pitch 392)",
                      "store b128 linear", "store b128 rows"};
  for (int wa : {0x01, 0x11, 0x03, 0x0f, 0xff}) {
    for (int p = 0; p < 6; p++) {
      if (p < 4) hipLaunchKernelGGL(k_store<8>, dim3(1), dim3(512), 65536, 0, p, d, wa);
      else hipLaunchKernelGGL(k_store<16>, dim3(1), dim3(512), 65536, 0, p, d, wa);
      report(sn[p], p < 4 ? 512 : 1024, wa);
    }
    const char* rn[] = {"tr read, fragment-order weights", "tr read, permuted slots", "tr read, hand-over image"};
    for (int q = 0; q < 3; q++) { hipLaunchKernelGGL(k_tr, dim3(1), dim3(512), 65536, 0, q, d, wa, sink); report(rn[q], 512, wa); }
    for (int ga = 0; ga < 2; ga++) {
      hipLaunchKernelGGL(k_read<4>, dim3(1), dim3(512), 65536, 0, d, wa, sink, ga); report(ga ? "read b32 gather (256 entries)" : "read b32 linear", 256, wa);
      hipLaunchKernelGGL(k_read<8>, dim3(1), dim3(512), 65536, 0, d, wa, sink, ga); report(ga ? "read b64 gather (256 entries)" : "read b64 linear", 512, wa);
      hipLaunchKernelGGL(k_read<16>, dim3(1), dim3(512), 65536, 0, d, wa, sink, ga); report(ga ? "read b128 gather (256 entries)" : "read b128 linear", 1024, wa);
    }
  }
  return 0;
}
