// MFMA + VALU co-issue microbenchmark (development aid).  One dependent accumulation chain of
// v_mfma_f32_32x32x16_bf16 per wave, with NV VALU instructions of a given kind pinned into every MFMA gap (asm
// volatile keeps program order).  Reports shader cycles per MFMA per SIMD (s_memtime) for 1 and 2 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// KIND 0: none, 1: v_fma x NV, 2: v_exp x NV, 3: the SiLU quintet (exp add rcp mul fmac) x NV/5 ... per gap
template <int KIND, int NV>
__global__ void __launch_bounds__(512, 1) k(float* out, long long* clk, int iters) {
  f32x16 acc = {0};
  u32x4 a = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}, b = a;
  float v[8];
  for (int i = 0; i < 8; i++) v[i] = 0.001f * (threadIdx.x + i);
  float t[8] = {0};
  const float c = 1.0000001f;
  long long t0 = clock64();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int k = 0; k < 16; k++) {
      asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
      if (KIND == 1) {
#pragma unroll
        for (int j = 0; j < NV; j++) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[j % 8]) : "v"(c));
      } else if (KIND == 2) {
#pragma unroll
        for (int j = 0; j < NV; j++) asm volatile("v_exp_f32 %0, %0" : "+v"(v[j % 8]));
      } else if (KIND == 3) {
#pragma unroll
        for (int j = 0; j < NV; j++) {  // one SiLU value pair half per MFMA, alternating halves as in the kernel
          if (k & 1)
            asm volatile("v_rcp_f32 %1, %1\n\tv_mul_f32 %0, %2, %0\n\tv_mul_f32 %1, %3, %1\n\tv_fmac_f32 %2, %4, %0\n\tv_fmac_f32 %3, %4, %1"
                         : "+v"(t[2 * j]), "+v"(t[2 * j + 1]), "+v"(v[2 * j]), "+v"(v[2 * j + 1]) : "s"(c));
          else
            asm volatile("v_exp_f32 %0, %2\n\tv_exp_f32 %1, %3\n\tv_add_f32 %0, 1.0, %0\n\tv_add_f32 %1, 1.0, %1\n\tv_rcp_f32 %0, %0"
                         : "+v"(t[2 * j]), "+v"(t[2 * j + 1]) : "v"(v[2 * j]), "v"(v[2 * j + 1]));
        }
      }
    }
  }
  long long t1 = clock64();
  float s = 0;
  for (int i = 0; i < 16; i++) s += acc[i];
  for (int i = 0; i < 8; i++) s += v[i] + t[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}

template <int KIND, int NV> void run(const char* name, float* out, long long* clk) {
  const int iters = 2000;
  for (int threads : {256, 512}) {  // 1 or 2 waves per SIMD (one workgroup per CU)
    hipLaunchKernelGGL((k<KIND, NV>), dim3(256), dim3(threads), 0, 0, out, clk, 10);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<KIND, NV>), dim3(256), dim3(threads), 0, 0, out, clk, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c;
    hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost);
    const double per_mfma_wave = (double)c / (iters * 16.0);
    const int wps = threads / 256;
    const double flops = 256.0 * (threads / 64) * iters * 16.0 * 32768.0;
    printf("%-28s waves/SIMD=%d  cycles per MFMA of one wave = %7.1f   => per SIMD-MFMA = %6.1f   host: %.3f ms = %.0f TFLOP/s, clk %.0f MHz\n",
           name, wps, per_mfma_wave, per_mfma_wave / wps, ms, flops / (ms * 1e-3) / 1e12, (double)c / (ms * 1e3));
  }
}

int main() {
  float* out; long long* clk;
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&clk, 8);
  run<0, 0>("mfma only", out, clk);
  run<1, 2>("mfma + 2 v_fma", out, clk);
  run<1, 5>("mfma + 5 v_fma", out, clk);
  run<1, 8>("mfma + 8 v_fma", out, clk);
  run<2, 1>("mfma + 1 v_exp", out, clk);
  run<2, 2>("mfma + 2 v_exp", out, clk);
  run<2, 4>("mfma + 4 v_exp", out, clk);
  run<3, 1>("mfma + silu half-pair (5 ops)", out, clk);
  return 0;
}
