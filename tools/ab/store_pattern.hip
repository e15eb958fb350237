// Does the register-dump store pattern of the wide network's stash (two 16-byte stores per lane into the lane's own 32 bytes: each
// instruction fills half of every cache line) cost HBM write bandwidth against 1-KiB-contiguous instructions?  (tools/ab: measurement)
//   hipcc -O3 --offload-arch=gfx950 tools/ab/store_pattern.hip -o build/store_pattern && build/store_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int PAT>
__global__ void __launch_bounds__(512, 1) k(char* out, long nblocks16k) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (long g = blockIdx.x; g * 8 + wave < nblocks16k; g += gridDim.x) {
    char* blk = out + (g * 8 + wave) * 16384;
#pragma unroll
    for (int to = 0; to < 8; to++) {
      const u32x4 a = {(unsigned)g, (unsigned)to, (unsigned)lane, 1u}, b = {2u, 3u, (unsigned)lane, (unsigned)g};
      if (PAT == 0) {  // the stash's: lane's own 32 bytes, two stores
        u32x4* d = reinterpret_cast<u32x4*>(blk + to * 2048 + (2 * (lane & 31) + (lane >> 5)) * 32);
        d[0] = a; d[1] = b;
      } else if (PAT == 1) {  // each store instruction one contiguous KiB
        u32x4* d = reinterpret_cast<u32x4*>(blk + to * 2048 + lane * 16);
        d[0] = a; d[64] = b;
      } else {  // non-temporal, stash pattern
        u32x4* d = reinterpret_cast<u32x4*>(blk + to * 2048 + (2 * (lane & 31) + (lane >> 5)) * 32);
        __builtin_nontemporal_store(a, d); __builtin_nontemporal_store(b, d + 1);
      }
    }
  }
}
int main() {
  const long bytes = 3L << 30, nb = bytes / 16384;
  char* d; hipMalloc(&d, bytes);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const char* names[] = {"lane-own 32 B, two stores (the stash)", "1 KiB contiguous per store", "the stash pattern, non-temporal"};
  for (int rep = 0; rep < 2; rep++)
    for (int p = 0; p < 3; p++) {
      hipEventRecord(e0);
      if (p == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(512), 0, 0, d, nb);
      if (p == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(512), 0, 0, d, nb);
      if (p == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(512), 0, 0, d, nb);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (rep) printf("%-44s %.3f ms  %.2f TB/s\n", names[p], ms, bytes / ms / 1e9);
    }
  return 0;
}
