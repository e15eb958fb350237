// tools/trace_resnet.hip -- development aid: per-phase clock64() trace of workgroup 0 of the wide-network forward.
// Build: hipcc -O3 --offload-arch=gfx950 -std=c++17 -fno-slp-vectorize -DSO3X_TRACE -o build/trace_resnet tools/trace_resnet.hip \
//        -Ldiffusion-extensions_amd -lso3x -Wl,-rpath,'$ORIGIN/../diffusion-extensions_amd'
// The exported so3x_resnet_* of this translation unit shadow the library's; everything else comes from libso3x.so.
#include "../diffusion-extensions_amd/csrc/so3x_resnet.hip"
#include <stdio.h>
#include <vector>

__global__ void k_clk(long long* o) { o[0] = clock64(); o[1] = wall_clock64(); }

int main(int argc, char** argv) {
  const int64_t n = argc > 1 ? atoll(argv[1]) : 65536;
  const int T = 1000;
  std::vector<float> hp(SO3X_RESNET_PARAMS), hx(n * 9, 0.f);
  for (size_t i = 0; i < hp.size(); i++) hp[i] = 0.06f * ((float)((i * 2654435761u) % 2001) / 1000.f - 1.f);
  for (int64_t i = 0; i < n; i++) hx[i * 9] = hx[i * 9 + 4] = hx[i * 9 + 8] = 1.f;
  std::vector<int64_t> ht(n);
  for (int64_t i = 0; i < n; i++) ht[i] = (i * 7) % T;
  float *p, *x, *out; int64_t* t; void* ws;
  const size_t wsb = so3x_resnet_workspace_bytes(SO3X_PREC_BF16, T);
  hipMalloc(&p, hp.size() * 4); hipMalloc(&x, hx.size() * 4); hipMalloc(&out, n * 12); hipMalloc(&t, n * 8); hipMalloc(&ws, wsb);
  hipMemcpy(p, hp.data(), hp.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(t, ht.data(), n * 8, hipMemcpyHostToDevice);
  for (int rep = 0; rep < 3; rep++) {
    int zero[8] = {0};
    hipMemcpyToSymbol(HIP_SYMBOL(g_trace_n), zero, sizeof(zero));
    int rc = so3x_resnet_fwd(nullptr, p, x, t, 1, out, n, SO3X_PREC_BF16, T, ws, wsb);
    hipDeviceSynchronize();
    if (rc) { printf("rc=%d\n", rc); return 1; }
  }
  {  // shader clock under this load: clock64 (shader cycles) vs wall_clock64 (100 MHz) across a long launch
    const int64_t nb = 1 << 20;
    float *xb, *ob; int64_t* tb;
    hipMalloc(&xb, nb * 36); hipMalloc(&ob, nb * 12); hipMalloc(&tb, nb * 8);
    hipMemset(xb, 0, nb * 36); hipMemset(tb, 0, nb * 8);
    long long* clk; hipMalloc(&clk, 32);
    k_clk<<<1, 1>>>(clk);
    for (int r = 0; r < 20; r++) so3x_resnet_fwd(nullptr, p, xb, tb, 1, ob, nb, SO3X_PREC_BF16, T, ws, wsb);
    k_clk<<<1, 1>>>(clk + 2);
    hipDeviceSynchronize();
    long long h[4]; hipMemcpy(h, clk, 32, hipMemcpyDeviceToHost);
    printf("shader clock under load: %.1f MHz (%.2f ms)\n", 100.0 * (double)(h[2] - h[0]) / (double)(h[3] - h[1]), (h[3] - h[1]) / 1e5);
  }
  static long long tr[8][1024]; int cnt[8];
  hipMemcpyFromSymbol(tr, HIP_SYMBOL(g_trace), sizeof(tr));
  hipMemcpyFromSymbol(cnt, HIP_SYMBOL(g_trace_n), sizeof(cnt));
  // print per wave: chunk index, then the deltas between consecutive stamps
  for (int w : {0, 4, 1, 5}) {
    printf("wave %d (%d stamps): id:delta ...\n", w, cnt[w]);
    long long prev = tr[w][0] >> 4;
    for (int i = 0; i < cnt[w] && i < 5 * 20; i++) {
      long long c = tr[w][i] >> 4; int id = (int)(tr[w][i] & 15);
      printf("%d:%lld ", id, c - prev); prev = c;
      if (id == 4) printf("| ");
      if (i % 20 == 19) printf("\n");
    }
    printf("\n");
  }
  // average per-chunk period over the first pass
  for (int w = 0; w < 8; w++) {
    long long first = -1, last = -1; int k = 0;
    for (int i = 0; i < cnt[w]; i++) if ((tr[w][i] & 15) == 1) { if (first < 0) first = tr[w][i] >> 4; last = tr[w][i] >> 4; k++; }
    printf("wave %d: %d barriers, mean period %.1f clk\n", w, k, k > 1 ? (double)(last - first) / (k - 1) : 0.0);
  }
  return 0;
}
