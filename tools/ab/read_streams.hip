// HBM read rate by how the workgroups divide the buffer: 252 workgroups x 512 threads, 16-KiB blocks, each workgroup either a
// CONTIGUOUS range (the wide network's dW kernel: one range of X_l and one of dZ_l per workgroup) or every 252nd block  (measurement)
#include <hip/hip_runtime.h>
#include <cstdio>
template <int STRIDED>
__global__ void __launch_bounds__(512, 1) k(const float4* __restrict__ a, const float4* __restrict__ b, long nblk, float* out) {
  float4 acc = {0, 0, 0, 0};
  const long per = (nblk + gridDim.x - 1) / gridDim.x;
  for (long i = 0; i < per; i++) {
    const long blk = STRIDED ? i * gridDim.x + blockIdx.x : blockIdx.x * per + i;
    if (blk >= nblk) break;
    const float4* pa = a + blk * 1024 + threadIdx.x;
    const float4* pb = b + blk * 1024 + threadIdx.x;
    const float4 v0 = pa[0], v1 = pa[512], w0 = pb[0], w1 = pb[512];
    acc.x += v0.x + v1.y + w0.z + w1.w;
  }
  if (acc.x == 12345.678f) out[0] = acc.x;
}
int main() {
  const long bytes = 2L << 30, nblk = bytes / 16384;
  float4 *a, *b; float* o; hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&o, 4); hipMemset(a, 0, bytes); hipMemset(b, 0, bytes);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 2; rep++)
    for (int s = 0; s < 2; s++) {
      hipEventRecord(e0);
      if (s) hipLaunchKernelGGL(k<1>, dim3(252), dim3(512), 0, 0, a, b, nblk, o); else hipLaunchKernelGGL(k<0>, dim3(252), dim3(512), 0, 0, a, b, nblk, o);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (rep) printf("%s  %.3f ms  %.2f TB/s\n", s ? "every 252nd block   " : "contiguous per group", ms, 2.0 * bytes / ms / 1e9);
    }
  return 0;
}
