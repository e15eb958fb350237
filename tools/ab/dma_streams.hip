// The wide network's dW kernel without its arithmetic: 252 workgroups x 8 waves stream [X | dZ] block pairs (2 x 16 KiB) into a
// four-slot LDS ring by LDS-DMA, three pairs in flight, one raw barrier per pair -- the rate the LOADER alone reaches, against the
// 5.9 TB/s of plain loads (tools/ab/read_streams.hip).  WORK = the transposed reads of the real kernel added (no MFMAs).  (measurement)
#include <hip/hip_runtime.h>
#include <cstdio>
template <int KEEP> __device__ __forceinline__ void ring_sync() { asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(KEEP) : "memory"); }
template <int WORK>
__global__ void __launch_bounds__(512, 1) k(const char* __restrict__ xs, const char* __restrict__ ds, long nblk, float* out) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long per = (nblk + gridDim.x - 1) / gridDim.x, b0 = blockIdx.x * per, b1 = b0 + per < nblk ? b0 + per : nblk;
  auto issue = [&](long blk, int slot) {
    const char* gx = xs + blk * 16384 + (2 * wave) * 1024 + lane * 16;
    const char* gd = ds + blk * 16384 + (2 * wave) * 1024 + lane * 16;
    char* lx = lds + slot * 32768 + (2 * wave) * 1024;
#pragma unroll
    for (int i = 0; i < 2; i++) {
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gx + i * 1024), (__attribute__((address_space(3))) void*)(lx + i * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gd + i * 1024), (__attribute__((address_space(3))) void*)(lx + 16384 + i * 1024), 16, 0, 0);
    }
  };
  for (int u = 0; u < 3; u++) if (b0 + u < b1) issue(b0 + u, u);
  int slot = 0; float acc = 0;
  typedef short s16x4 __attribute__((ext_vector_type(4)));
  typedef __attribute__((address_space(3))) s16x4* lds_p;
  for (long blk = b0; blk < b1; blk++) {
    const long younger = b1 - 1 - blk;
    if (younger >= 2) ring_sync<8>(); else if (younger == 1) ring_sync<4>(); else ring_sync<0>();
    if (blk + 3 < b1) issue(blk + 3, (slot + 3) & 3);
    if (WORK) {
      const char* img = lds + slot * 32768;
#pragma unroll
      for (int i = 0; i < 36; i++) { const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(img + (i * 896 + lane * 8) % 32000)); acc += v[0]; }
    }
    slot = (slot + 1) & 3;
  }
  if (acc == 12345.678f) out[0] = acc;
}
int main() {
  const long bytes = 2L << 30, nblk = bytes / 16384;
  char *a, *b; float* o; hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&o, 4); hipMemset(a, 0, bytes); hipMemset(b, 0, bytes);
  hipFuncSetAttribute((const void*)k<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  hipFuncSetAttribute((const void*)k<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 2; rep++)
    for (int w = 0; w < 2; w++) {
      hipEventRecord(e0);
      if (w) hipLaunchKernelGGL(k<1>, dim3(252), dim3(512), 131072, 0, a, b, nblk, o); else hipLaunchKernelGGL(k<0>, dim3(252), dim3(512), 131072, 0, a, b, nblk, o);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (rep) printf("%s  %.3f ms  %.2f TB/s\n", w ? "loader + 36 transposed reads per wave and pair" : "loader alone                                 ", ms, 2.0 * bytes / ms / 1e9);
    }
  return 0;
}
