// VALU issue-rate microbenchmark (development aid): how many cycles does a wave64 instruction
// cost on one SIMD as a function of waves/SIMD?  Prints wave-instructions per cycle per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int OP>
__global__ void __launch_bounds__(256) k(float* out, int iters, float seed) {
  float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  const float m = 1.0000001f, c = 1e-9f;
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
  const f2 pm = {m, m}, pc = {c, c};
  for (int i = 0; i < iters; i++) {
    if (OP == 0) {  // v_fma_f32 x8 independent
      a0 = fmaf(a0, m, c); a1 = fmaf(a1, m, c); a2 = fmaf(a2, m, c); a3 = fmaf(a3, m, c);
      a4 = fmaf(a4, m, c); a5 = fmaf(a5, m, c); a6 = fmaf(a6, m, c); a7 = fmaf(a7, m, c);
    } else if (OP == 1) {  // v_pk_fma_f32 x4 (8 fmas)
      p0 = __builtin_elementwise_fma(p0, pm, pc); p1 = __builtin_elementwise_fma(p1, pm, pc);
      p2 = __builtin_elementwise_fma(p2, pm, pc); p3 = __builtin_elementwise_fma(p3, pm, pc);
    } else if (OP == 2) {  // v_exp_f32 x8
      a0 = __builtin_amdgcn_exp2f(a0); a1 = __builtin_amdgcn_exp2f(a1); a2 = __builtin_amdgcn_exp2f(a2); a3 = __builtin_amdgcn_exp2f(a3);
      a4 = __builtin_amdgcn_exp2f(a4); a5 = __builtin_amdgcn_exp2f(a5); a6 = __builtin_amdgcn_exp2f(a6); a7 = __builtin_amdgcn_exp2f(a7);
    } else if (OP == 3) {  // v_rcp_f32 x8
      a0 = __builtin_amdgcn_rcpf(a0); a1 = __builtin_amdgcn_rcpf(a1); a2 = __builtin_amdgcn_rcpf(a2); a3 = __builtin_amdgcn_rcpf(a3);
      a4 = __builtin_amdgcn_rcpf(a4); a5 = __builtin_amdgcn_rcpf(a5); a6 = __builtin_amdgcn_rcpf(a6); a7 = __builtin_amdgcn_rcpf(a7);
    } else if (OP == 4) {  // v_mul_f32 x8
      a0 *= m; a1 *= m; a2 *= m; a3 *= m; a4 *= m; a5 *= m; a6 *= m; a7 *= m;
    } else if (OP == 5) {  // v_cvt_pk_bf16_f32 x4 + unpack-ish
      typedef __bf16 b2 __attribute__((ext_vector_type(2)));
      b2 q0 = {(__bf16)a0, (__bf16)a1}, q1 = {(__bf16)a2, (__bf16)a3}, q2 = {(__bf16)a4, (__bf16)a5}, q3 = {(__bf16)a6, (__bf16)a7};
      a0 = (float)q0[0] + c; a1 = (float)q0[1] + c; a2 = (float)q1[0] + c; a3 = (float)q1[1] + c;
      a4 = (float)q2[0] + c; a5 = (float)q2[1] + c; a6 = (float)q3[0] + c; a7 = (float)q3[1] + c;
    } else if (OP == 6) {  // mix: silu-like exp2, add, rcp, mul on 4 values
      float t0 = __builtin_amdgcn_exp2f(a0), t1 = __builtin_amdgcn_exp2f(a1), t2 = __builtin_amdgcn_exp2f(a2), t3 = __builtin_amdgcn_exp2f(a3);
      t0 = __builtin_amdgcn_rcpf(1.0f + t0); t1 = __builtin_amdgcn_rcpf(1.0f + t1); t2 = __builtin_amdgcn_rcpf(1.0f + t2); t3 = __builtin_amdgcn_rcpf(1.0f + t3);
      a0 = a0 * t0 + c; a1 = a1 * t1 + c; a2 = a2 * t2 + c; a3 = a3 * t3 + c;
    }
  }
  if (OP == 1) { a0 = p0[0] + p0[1] + p1[0] + p1[1] + p2[0] + p2[1] + p3[0] + p3[1]; }
  out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

template <int OP> void run(const char* name, int insts_per_iter, float* out) {
  const int iters = 20000;
  for (int wps : {1, 2, 4, 8}) {
    int blocks = 256 * wps;  // 256 CUs x wps blocks of 4 waves = wps waves per SIMD
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, 100, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    double winst = (double)blocks * 4 * iters * insts_per_iter;   // wave-instructions
    double per_simd_per_s = winst / 1024.0 / (ms * 1e-3);
    printf("%-12s waves/SIMD=%d  %.3f ms  wave-inst/s/SIMD=%.3e  => cycles/inst @2.4GHz = %.2f\n", name, wps, ms, per_simd_per_s,
           2.4e9 / per_simd_per_s);
  }
}

int main() {
  float* out;
  hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
  run<0>("v_fma_f32", 8, out);
  run<1>("v_pk_fma_f32", 4, out);
  run<4>("v_mul_f32", 8, out);
  run<2>("v_exp_f32", 8, out);
  run<3>("v_rcp_f32", 8, out);
  run<5>("cvt_bf16 rt", 16, out);
  run<6>("silu4 (16)", 16, out);
  return 0;
}
