// Store-pattern microbenchmark (development aid): HBM write rate of a wave writing 32 B per lane as
//   A: two 16-byte stores at a 32-byte lane stride (each instruction covers half of every 32-byte segment), vs
//   B: two 16-byte stores that are each a contiguous 1 KiB across the wave.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ void __launch_bounds__(512) k(uint4* out, long long tiles_per_wave) {
  const int lane = threadIdx.x & 63;
  const long long wave = (long long)blockIdx.x * 8 + (threadIdx.x >> 6), nw = (long long)gridDim.x * 8;
  uint4 v = {(unsigned)lane, 1u, 2u, 3u};
  for (long long i = 0; i < tiles_per_wave; i++) {
    uint4* base = out + (i * nw + wave) * 128;  // 2 KiB per wave-tile
    if (MODE == 0) { base[2 * lane] = v; base[2 * lane + 1] = v; }
    else { base[lane] = v; base[64 + lane] = v; }
    v.y += 1;
  }
}
int main() {
  const long long bytes = 4ll << 30;
  uint4* buf; hipMalloc(&buf, bytes);
  const int blocks = 256;
  const long long tpw = bytes / 2048 / (blocks * 8);
  for (int mode = 0; mode < 2; mode++)
    for (int rep = 0; rep < 3; rep++) {
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      hipEventRecord(e0);
      if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(512), 0, 0, buf, tpw);
      else hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(512), 0, 0, buf, tpw);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      printf("mode %c: %.3f ms  %.2f TB/s\n", mode ? 'B' : 'A', ms, bytes / (ms * 1e-3) / 1e12);
    }
  return 0;
}
