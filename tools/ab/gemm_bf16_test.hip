// Standalone check + timing of the bf16 GEMM kernels of so3x_planenet_bf16.hip against a naive device reference.
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -I diffusion-extensions_amd/csrc tools/ab/gemm_bf16_test.hip -o build/gemm_bf16_test
#include "../../diffusion-extensions_amd/csrc/so3x_planenet_bf16.hip"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
using namespace so3x::plane;
namespace so3x { namespace plane {   // the fp32 translation unit is not linked here
int gemm(hipStream_t, Mat, Mat, float*, int64_t, int, int, int, const float*, float, bool, bool, int, int, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t) { return 0; }
int head(hipStream_t, const float*, const float*, const float*, const float*, const float*, float*, float*, int64_t, int) { return 0; }
} }
__global__ void k_ref(const bf16* A, const bf16* W, float* C, const float* bias, const bf16* R, int M, int N, int K, int relu) {
  const int n = blockIdx.x * 16 + (threadIdx.x & 15), m = blockIdx.y * 16 + (threadIdx.x >> 4);
  float acc = 0.f;
  for (int k = 0; k < K; k++) acc += (float)A[(size_t)m * K + k] * (float)W[(size_t)n * K + k];
  acc += bias[n];
  if (R && relu != 3) acc += (float)R[(size_t)m * N + n];
  if (relu == 1) acc = fmaxf(acc, 0.f);
  if (relu == 3) acc = (float)R[(size_t)m * N + n] > 0.f ? acc : 0.f;
  C[(size_t)m * N + n] = acc;
}
__global__ void k_fill(bf16* p, size_t n, unsigned seed) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  unsigned x = (unsigned)i * 2654435761u + seed;
  x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
  p[i] = (bf16)(((int)(x & 0xffff) - 32768) / 32768.0f);
}
int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 65536, N = argc > 2 ? atoi(argv[2]) : 1536, K = argc > 3 ? atoi(argv[3]) : 512;
  const int mode = argc > 4 ? atoi(argv[4]) : 0;   // 0 plain, 1 relu, 2 resid
  bf16 *A, *W, *C, *R; float *ref, *bias;
  hipMalloc(&A, (size_t)M * K * 2); hipMalloc(&W, (size_t)N * K * 2); hipMalloc(&C, (size_t)M * N * 2); hipMalloc(&R, (size_t)M * N * 2);
  hipMalloc(&ref, (size_t)M * N * 4); hipMalloc(&bias, N * 4);
  k_fill<<<(unsigned)(((size_t)M * K + 255) / 256), 256>>>(A, (size_t)M * K, 1);
  k_fill<<<(unsigned)(((size_t)N * K + 255) / 256), 256>>>(W, (size_t)N * K, 2);
  k_fill<<<(unsigned)(((size_t)M * N + 255) / 256), 256>>>(R, (size_t)M * N, 3);
  std::vector<float> hb(N);
  for (int i = 0; i < N; i++) hb[i] = 0.01f * (i % 37);
  hipMemcpy(bias, hb.data(), N * 4, hipMemcpyHostToDevice);
  k_ref<<<dim3(N / 16, M / 16), 256>>>(A, W, ref, bias, mode >= 2 ? R : nullptr, M, N, K, mode);
  hipMemset(C, 0xff, (size_t)M * N * 2);
  int rc = gemm_bf16(0, A, K, W, K, C, N, bias, mode >= 2 ? R : nullptr, N, M, N, K, mode);
  hipDeviceSynchronize();
  printf("rc %d err %s\n", rc, hipGetErrorString(hipGetLastError()));
  std::vector<unsigned short> hc((size_t)M * N);
  std::vector<float> hr((size_t)M * N);
  hipMemcpy(hc.data(), C, hc.size() * 2, hipMemcpyDeviceToHost);
  hipMemcpy(hr.data(), ref, hr.size() * 4, hipMemcpyDeviceToHost);
  double maxerr = 0; size_t bad = 0; int printed = 0;
  std::vector<int> badrow(256, 0), badcol(256, 0);
  for (size_t i = 0; i < hc.size(); i++) {
    unsigned u = (unsigned)hc[i] << 16; float v; memcpy(&v, &u, 4);
    const double e = fabs((double)v - hr[i]), tol = 0.02 + 0.01 * fabs(hr[i]);
    if (!(e <= tol)) {
      bad++; badrow[(i / N) % 256]++; badcol[(i % N) % 256]++;
      if (printed++ < 6) printf("  bad at m %zu n %zu: got %g want %g\n", i / N, i % N, v, hr[i]);
    }
    if (e > maxerr) maxerr = e;
  }
  printf("M %d N %d K %d mode %d: max err %g, bad %zu of %zu\n", M, N, K, mode, maxerr, bad, hc.size());
  if (bad) {
    printf("bad rows%%256:"); for (int i = 0; i < 256; i++) if (badrow[i]) printf(" %d", i); printf("\nbad cols%%256:");
    for (int i = 0; i < 256; i++) if (badcol[i]) printf(" %d", i); printf("\n");
  }
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 5; i++) gemm_bf16(0, A, K, W, K, C, N, bias, mode >= 2 ? R : nullptr, N, M, N, K, mode);
  hipEventRecord(e0);
  const int reps = 20;
  for (int i = 0; i < reps; i++) gemm_bf16(0, A, K, W, K, C, N, bias, mode >= 2 ? R : nullptr, N, M, N, K, mode);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
  printf("time %.1f us  %.0f TFLOP/s (%.1f %% of 2.5 PF)\n", ms * 1e3, 2.0 * M * N * K / ms / 1e9, 2.0 * M * N * K / ms / 1e9 / 25.0);
  return bad ? 1 : 0;
}
