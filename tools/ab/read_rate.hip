// practical HBM read rate: 256 x 4 workgroups streaming 4 GiB with 16-byte loads, 1/2/4/8 loads in flight per lane  (tools/ab: measurement)
#include <hip/hip_runtime.h>
#include <cstdio>
template <int U>
__global__ void __launch_bounds__(256) k(const float4* __restrict__ in, long n4, float* out) {
  float4 acc = {0, 0, 0, 0};
  const long stride = (long)gridDim.x * 256;
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  for (; i + (U - 1) * stride < n4; i += U * stride) {
    float4 v[U];
#pragma unroll
    for (int u = 0; u < U; u++) v[u] = in[i + u * stride];
#pragma unroll
    for (int u = 0; u < U; u++) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
  }
  if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = acc.x;
}
int main() {
  const long bytes = 4L << 30, n4 = bytes / 16;
  float4* d; float* o; hipMalloc(&d, bytes); hipMalloc(&o, 4); hipMemset(d, 0, bytes);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int grid : {1024, 2048, 8192})
    for (int u : {1, 2, 4, 8}) {
      float best = 1e9;
      for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0);
        if (u == 1) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 0, 0, d, n4, o);
        if (u == 2) hipLaunchKernelGGL(k<2>, dim3(grid), dim3(256), 0, 0, d, n4, o);
        if (u == 4) hipLaunchKernelGGL(k<4>, dim3(grid), dim3(256), 0, 0, d, n4, o);
        if (u == 8) hipLaunchKernelGGL(k<8>, dim3(grid), dim3(256), 0, 0, d, n4, o);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best;
      }
      printf("grid %5d  loads in flight per lane %d   %.3f ms   %.2f TB/s\n", grid, u, best, bytes / best / 1e9);
    }
  return 0;
}
