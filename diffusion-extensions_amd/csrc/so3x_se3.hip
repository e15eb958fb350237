// so3x_se3.hip -- SE(3) = SO(3) x R^3 layer: IGSO3xR3 noise
// (distributions.py:84-110), SE3Diffusion forward noising / targets / reverse mean / reverse
// noise (diffusion.py:432-522) and the rigid move of a residue set (prot_util.py:73-81).
// The rotation part reuses the SO(3) device math; the shift part is a scaled Gaussian.
#include "so3x_common.hpp"
#include "so3x_math.hpp"
#include "so3x_igso3.hpp"

using namespace so3x;

namespace {

enum { S_SQRT_AC = 3, S_SQRT_1MAC = 4, S_RECIP = 6, S_RECIPM1 = 7, S_COEF1 = 10, S_COEF2 = 11 };
constexpr uint64_t kNormalStream = 1ull << 62;  // Philox counter plane of the shift normals

__device__ __forceinline__ void unit_axis_from(const float* a, float* ax) {  // distributions.py:36 + util.py:201
  float nrm = sqrtf(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]);
  ax[0] = a[0] / nrm; ax[1] = a[1] / nrm; ax[2] = a[2] / nrm;
  float n2 = sqrtf(ax[0] * ax[0] + ax[1] * ax[1] + ax[2] * ax[2]);
  ax[0] /= n2; ax[1] /= n2; ax[2] /= n2;
}

// q_sample + p_losses targets: 84 B (rotation) + 24 B in / 24 B out (shift) per sample.
// Round 3: laid out like k_q_sample_target (so3x_diffusion.hip) -- wave-private staging, one 64-frame tile per wave on an
// oversubscribed grid (no workgroup barrier: the per-frame CDF-row search is a chain of dependent L2 round trips that only other
// resident waves hide), and with in-kernel draws (LEAN) the frame's rotation tile comes in by LDS-DMA while the noise chain runs
// and the rotation target is axis * angle itself.  timesteps are clamped to [0, T-1] as everywhere.
template <bool LEAN>
__global__ void __launch_bounds__(kBlock, LEAN ? 6 : 4)
k_se3_q_sample_target(const float* __restrict__ sched, int T, const float* __restrict__ trap_q,
                      const uint16_t* __restrict__ guide_q, float shift_scale,
                      const float* __restrict__ x0_rot, const float* __restrict__ x0_shift, const int64_t* __restrict__ t,
                      int quirk_col0, const float* __restrict__ axes_, const float* __restrict__ unif_,
                      const float* __restrict__ znorm_, uint64_t seed, uint64_t rng_offset, int64_t index_base,
                      float* __restrict__ xt_rot, float* __restrict__ xt_shift, float* __restrict__ target_rot,
                      float* __restrict__ target_shift, int64_t n) {
  __shared__ __attribute__((aligned(16))) float sm[kBlock / kWave][kWave * 9];
  const float* axes = LEAN ? nullptr : axes_;
  const float* unif = LEAN ? nullptr : unif_;
  const float* znorm = LEAN ? nullptr : znorm_;
  float* wl = sm[threadIdx.x >> 6];
  const int lane = threadIdx.x & 63;
  const int64_t ntiles = (n + kWave - 1) / kWave;
  const int64_t wave = (int64_t)blockIdx.x * (kBlock / kWave) + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * (kBlock / kWave);
  auto clamp_t = [&](int64_t v) -> int64_t { return v < 0 ? 0 : (v >= T ? T - 1 : v); };
  const int64_t wrow_t = quirk_col0 ? clamp_t(t[0]) : -1;
  for (int64_t tile = wave; tile < ntiles; tile += nwaves) {
    const int64_t base = tile * kWave;
    const int cnt = (int)((n - base) < kWave ? (n - base) : kWave);
    const int64_t idx = base + lane;
    const bool live = lane < cnt;
    const bool prefetched = LEAN && cnt == kWave && ((reinterpret_cast<uintptr_t>(x0_rot + base * 9) & 15) == 0);
    if (prefetched) wave_dma9(x0_rot, base, wl);
    const int64_t tt = clamp_t(t[live ? idx : base]);
    float ax[3], u, z[3];
    if (axes) {
      float a[3];
      wave_load_rows<3>(axes, base, cnt, wl, a);
      unit_axis_from(a, ax);
      u = live ? unif[idx] : 0.5f;
      wave_load_rows<3>(znorm, base, cnt, wl, z);
    } else {
      Philox4 r = philox4x32_10(seed, (uint64_t)(index_base + idx), rng_offset);
      unit_axis(r.x, r.y, ax);
      u = u01(r.z);
      Philox4 q = philox4x32_10(seed, (uint64_t)(index_base + idx), rng_offset | kNormalStream);
      float spare;
      box_muller(q.x, q.y, &z[0], &z[1]);
      box_muller(q.z, q.w, &z[2], &spare);
    }
    const float* row = trap_q + tt * 999;
    const float* wrow = wrow_t >= 0 ? trap_q + wrow_t * 999 : row;
    const float ang = igso3_angle_global(row, wrow, SO3X_KNOTS_DATA, u, guide_q ? guide_q + tt * kGuidePitch : nullptr);
    float nz[9], x[9], w[3], xs[9], xt[9], sh[3];
    exp_axis_angle(ax, ang, nz);
    if (prefetched) wave_dma9_commit(wl, x);
    else wave_load_rows<9>(x0_rot, base, cnt, wl, x);
    wave_load_rows<3>(x0_shift, base, cnt, wl, sh);
    const float k = sched[S_SQRT_AC * T + tt], eps = sched[S_SQRT_1MAC * T + tt];
    log3(x, w);
    w[0] *= k; w[1] *= k; w[2] *= k;
    exp3(w, xs);
    mul33(xs, nz, xt);                                   // x_blend.rot @ noise.rot   (diffusion.py:503)
    const float ns = eps * shift_scale;                  // Normal(scale = eps * shift_scale), distributions.py:96
    float xo[3] = {sh[0] * k + z[0] * ns, sh[1] * k + z[1] * ns, sh[2] * k + z[2] * ns};  // x_blend.shift + noise.shift
    if (xt_rot) wave_store_rows<9>(xt_rot, base, cnt, wl, xt);
    if (xt_shift) wave_store_rows<3>(xt_shift, base, cnt, wl, xo);
    if (target_rot) {
      float lw[3];
      if constexpr (LEAN) { lw[0] = ax[0] * ang; lw[1] = ax[1] * ang; lw[2] = ax[2] * ang; }  // log(exp(hat(axis * angle))), exactly
      else log3(nz, lw);
      const float ie = 1.0f / eps;
      float tg[3] = {lw[0] * ie, lw[1] * ie, lw[2] * ie};                               // diffusion.py:512
      wave_store_rows<3>(target_rot, base, cnt, wl, tg);
    }
    if (target_shift) {
      const float inv = 1.0f / (eps * shift_scale);                                      // diffusion.py:511
      float tg[3] = {z[0] * ns * inv, z[1] * ns * inv, z[2] * ns * inv};
      wave_store_rows<3>(target_shift, base, cnt, wl, tg);
    }
  }
}

// predict_start_from_noise + q_posterior for AffineT / AffineGrad (diffusion.py:444-466)
__global__ void __launch_bounds__(kBlock)
k_se3_p_mean(const float* __restrict__ sched, int T, const float* __restrict__ x_rot, const float* __restrict__ x_shift,
             const float* __restrict__ v_rot, const float* __restrict__ v_shift, int t, float* __restrict__ mean_rot,
             float* __restrict__ mean_shift, int64_t n) {
  __shared__ __attribute__((aligned(16))) float sm[kTile * 9];
  const float a = sched[S_RECIP * T + t], b = sched[S_RECIPM1 * T + t], c1 = sched[S_COEF1 * T + t], c2 = sched[S_COEF2 * T + t];
  const int64_t ntiles = (n + kTile - 1) / kTile;
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t base = tile * kTile;
    const int cnt = (int)((n - base) < kTile ? (n - base) : kTile);
    float xr[9], vr[3], xs[3], vs[3], xh[9], mr[9];
    load_rows<9>(x_rot, base, cnt, sm, xr);
    load_rows<3>(v_rot, base, cnt, sm, vr);
    load_rows<3>(x_shift, base, cnt, sm, xs);
    load_rows<3>(v_shift, base, cnt, sm, vs);
    p_mean_one(xr, vr, a, b, c1, c2, xh, mr);
    float ms[3];
#pragma unroll
    for (int j = 0; j < 3; j++) {
      const float x0h = xs[j] * a - vs[j] * b;   // x_t_term.shift - noise_shift   (:454)
      ms[j] = x0h * c1 + xs[j] * c2;             // c_1.shift + c_2.shift          (:459)
    }
    store_rows<9>(mean_rot, base, cnt, sm, mr);
    store_rows<3>(mean_shift, base, cnt, sm, ms);
  }
}

// p_sample noise (diffusion.py:476-483): IGSO3xR3(eps = sigma (scalar), mean = model_mean).sample() draws ONE
// rotation noise for the whole batch (scalar eps, empty sample shape; distributions.py:98-101) and independent
// Gaussian shifts.  shared_rot != 0 reproduces that; 0 draws one rotation noise per sample.
__global__ void __launch_bounds__(kBlock)
k_se3_p_noise(const float* __restrict__ trap_row, float sigma, float shift_scale, const float* __restrict__ mean_rot,
              const float* __restrict__ mean_shift, const float* __restrict__ axes, const float* __restrict__ unif,
              const float* __restrict__ znorm, uint64_t seed, uint64_t rng_offset, int64_t index_base, int shared_rot,
              float* __restrict__ out_rot, float* __restrict__ out_shift, int64_t n) {
  __shared__ __attribute__((aligned(16))) float sm[kTile * 9];
  const int64_t ntiles = (n + kTile - 1) / kTile;
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t base = tile * kTile;
    const int cnt = (int)((n - base) < kTile ? (n - base) : kTile);
    const int64_t idx = base + threadIdx.x;
    float ax[3], u, z[3];
    if (axes) {
      float a[3];
      if (shared_rot) { a[0] = axes[0]; a[1] = axes[1]; a[2] = axes[2]; u = unif[0]; }
      else { load_rows<3>(axes, base, cnt, sm, a); u = threadIdx.x < cnt ? unif[idx] : 0.5f; }
      unit_axis_from(a, ax);
      load_rows<3>(znorm, base, cnt, sm, z);
    } else {
      Philox4 r = philox4x32_10(seed, shared_rot ? 0ull : (uint64_t)(index_base + idx), rng_offset);
      unit_axis(r.x, r.y, ax);
      u = u01(r.z);
      Philox4 q = philox4x32_10(seed, (uint64_t)(index_base + idx), rng_offset | kNormalStream);
      float spare;
      box_muller(q.x, q.y, &z[0], &z[1]);
      box_muller(q.z, q.w, &z[2], &spare);
    }
    const float ang = igso3_angle(trap_row, trap_row, SO3X_KNOTS_DATA, u);
    float nz[9], mr[9], ms[3], o[9];
    exp_axis_angle(ax, ang, nz);
    load_rows<9>(mean_rot, base, cnt, sm, mr);
    load_rows<3>(mean_shift, base, cnt, sm, ms);
    mul33(mr, nz, o);                                   // mean.rot @ noise (distributions.py:50 with mean = model_mean.rot)
    const float ns = sigma * shift_scale;
    float os[3] = {ms[0] + z[0] * ns, ms[1] + z[1] * ns, ms[2] + z[2] * ns};
    store_rows<9>(out_rot, base, cnt, sm, o);
    store_rows<3>(out_shift, base, cnt, sm, os);
  }
}

// move_prot (prot_util.py:73-81): one workgroup per structure; mean over its L residue positions, then
//   pos' = (pos - mean) R^T + mean + shift,  frames' = frames R^T.   48 B in + 48 B out per residue.
__global__ void __launch_bounds__(kBlock)
k_rigid_move(const float* __restrict__ rot, const float* __restrict__ shift, const float* __restrict__ pos,
             const float* __restrict__ frames, float* __restrict__ out_pos, float* __restrict__ out_frames, int64_t L,
             const int64_t* __restrict__ off) {   // off != nullptr: ragged structures, structure s = rows off[s] .. off[s + 1]
  __shared__ float red[3][kBlock / 64];
  __shared__ float mean_s[3];
  const int64_t s = blockIdx.x;
  const int64_t row0 = off ? off[s] : s * L;
  if (off) L = off[s + 1] - row0;
  const float* p = pos + row0 * 3;
  float acc[3] = {0.f, 0.f, 0.f};
  for (int64_t i = threadIdx.x; i < L; i += kBlock) { acc[0] += p[i * 3]; acc[1] += p[i * 3 + 1]; acc[2] += p[i * 3 + 2]; }
#pragma unroll
  for (int j = 0; j < 3; j++) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) acc[j] += __shfl_down(acc[j], d);
    if ((threadIdx.x & 63) == 0) red[j][threadIdx.x >> 6] = acc[j];
  }
  __syncthreads();
  if (threadIdx.x < 3) {
    float m = 0.f;
    for (int w = 0; w < kBlock / 64; w++) m += red[threadIdx.x][w];
    mean_s[threadIdx.x] = m / (float)L;
  }
  __syncthreads();
  float R[9], sh[3], mu[3] = {mean_s[0], mean_s[1], mean_s[2]};
#pragma unroll
  for (int j = 0; j < 9; j++) R[j] = rot[s * 9 + j];
#pragma unroll
  for (int j = 0; j < 3; j++) sh[j] = shift[s * 3 + j];
  for (int64_t i = threadIdx.x; i < L; i += kBlock) {
    const float d0 = p[i * 3] - mu[0], d1 = p[i * 3 + 1] - mu[1], d2 = p[i * 3 + 2] - mu[2];
    float* o = out_pos + (row0 + i) * 3;
#pragma unroll
    for (int j = 0; j < 3; j++) o[j] = d0 * R[3 * j] + d1 * R[3 * j + 1] + d2 * R[3 * j + 2] + mu[j] + sh[j];  // (p - mu) @ R^T
    if (frames) {
      float f[9], fo[9];
      load_rot9(frames, row0 + i, f);
      mul33_bt(f, R, fo);
      store_rot9(out_frames, row0 + i, fo);
    }
  }
}

// PointCloudProj (models.py:75-91, so3 branch): out[b][p] = data[p] @ R_b^T -- one cloud shared by the whole batch,
// 12 B written per (rotation, point), the cloud and the rotation stay in registers / L1.  One workgroup per rotation
// and 768-point slab; consecutive threads write consecutive floats.
__global__ void __launch_bounds__(kBlock)
k_rotate_cloud(const float* __restrict__ rot, const float* __restrict__ cloud, int64_t cloud_stride, float* __restrict__ out, int64_t P) {
  const int64_t b = blockIdx.y;
  cloud += b * cloud_stride;  // 0: one cloud for every rotation; 3 P: a cloud per rotation (aircraft_rotate.py:104-106)
  float R[9];
#pragma unroll
  for (int j = 0; j < 9; j++) R[j] = rot[b * 9 + j];
  const int64_t f0 = (int64_t)blockIdx.x * (3 * kBlock);
#pragma unroll
  for (int k = 0; k < 3; k++) {
    const int64_t f = f0 + k * kBlock + threadIdx.x;  // flat index into [P][3]
    if (f < 3 * P) {
      const int64_t pt = f / 3;
      const int j = (int)(f - 3 * pt);
      const float d0 = cloud[pt * 3], d1 = cloud[pt * 3 + 1], d2 = cloud[pt * 3 + 2];
      out[b * 3 * P + f] = d0 * R[3 * j] + d1 * R[3 * j + 1] + d2 * R[3 * j + 2];
    }
  }
}

}  // namespace

extern "C" {

int so3x_rotate_cloud(so3x_stream_t s, const float* rot, const float* cloud, int64_t cloud_stride, float* out, int64_t n, int64_t P) {
  if (n < 0 || P < 0 || n > 65535 * (int64_t)65535 || ((n && P) && (!rot || !cloud || !out)) || (cloud_stride != 0 && cloud_stride != 3 * P))
    return SO3X_ERR_INVALID_ARG;
  if (n == 0 || P == 0) return SO3X_OK;
  const unsigned gx = (unsigned)((3 * P + 3 * kBlock - 1) / (3 * kBlock));
  for (int64_t b0 = 0; b0 < n; b0 += 65535) {  // grid.y limit
    const int64_t nb = n - b0 < 65535 ? n - b0 : 65535;
    hipLaunchKernelGGL(k_rotate_cloud, dim3(gx, (unsigned)nb), dim3(kBlock), 0, (hipStream_t)s, rot + b0 * 9, cloud + b0 * cloud_stride,
                       cloud_stride, out + b0 * 3 * P, P);
  }
  return check_launch();
}

int so3x_se3_q_sample_target(so3x_stream_t s, const float* sched, int T, const float* trap_q, const uint16_t* guide_q,
                             float shift_scale,
                             const float* x0_rot, const float* x0_shift, const int64_t* t, int quirk_col0, const float* axes,
                             const float* unif, const float* znorm, uint64_t seed, uint64_t rng_offset, int64_t index_base,
                             float* xt_rot, float* xt_shift, float* target_rot, float* target_shift, int64_t n) {
  const int nexp = (axes != nullptr) + (unif != nullptr) + (znorm != nullptr);
  if (n < 0 || T <= 0 || (n && (!sched || !trap_q || !x0_rot || !x0_shift || !t)) || (nexp != 0 && nexp != 3))
    return SO3X_ERR_INVALID_ARG;
  if (n == 0) return SO3X_OK;
  const int64_t nt64 = (n + kWave - 1) / kWave;
  int64_t want = (nt64 + 3) / 4;   // one tile per wave
  if (want > (1 << 20)) want = 1 << 20;
  if (nexp == 0)
    hipLaunchKernelGGL(k_se3_q_sample_target<true>, dim3((unsigned)want), dim3(kBlock), 0, (hipStream_t)s,
                       sched, T, trap_q, guide_q, shift_scale, x0_rot, x0_shift, t, quirk_col0, axes, unif, znorm, seed, rng_offset,
                       index_base, xt_rot, xt_shift, target_rot, target_shift, n);
  else
    hipLaunchKernelGGL(k_se3_q_sample_target<false>, dim3((unsigned)want), dim3(kBlock), 0, (hipStream_t)s,
                       sched, T, trap_q, guide_q, shift_scale, x0_rot, x0_shift, t, quirk_col0, axes, unif, znorm, seed, rng_offset,
                       index_base, xt_rot, xt_shift, target_rot, target_shift, n);
  return check_launch();
}

int so3x_se3_p_mean(so3x_stream_t s, const float* sched, int T, const float* x_rot, const float* x_shift, const float* v_rot,
                    const float* v_shift, int t, float* mean_rot, float* mean_shift, int64_t n) {
  if (n < 0 || T <= 0 || t < 0 || t >= T || (n && (!sched || !x_rot || !x_shift || !v_rot || !v_shift || !mean_rot || !mean_shift)))
    return SO3X_ERR_INVALID_ARG;
  if (n == 0) return SO3X_OK;
  hipLaunchKernelGGL(k_se3_p_mean, dim3(grid_for_tiles((n + kTile - 1) / kTile)), dim3(kBlock), 0, (hipStream_t)s, sched, T,
                     x_rot, x_shift, v_rot, v_shift, t, mean_rot, mean_shift, n);
  return check_launch();
}

int so3x_se3_p_noise(so3x_stream_t s, const float* trap_row, float sigma, float shift_scale, const float* mean_rot,
                     const float* mean_shift, const float* axes, const float* unif, const float* znorm, uint64_t seed,
                     uint64_t rng_offset, int64_t index_base, int shared_rot, float* out_rot, float* out_shift, int64_t n) {
  const int nexp = (axes != nullptr) + (unif != nullptr) + (znorm != nullptr);
  if (n < 0 || (n && (!trap_row || !mean_rot || !mean_shift || !out_rot || !out_shift)) || (nexp != 0 && nexp != 3))
    return SO3X_ERR_INVALID_ARG;
  if (n == 0) return SO3X_OK;
  hipLaunchKernelGGL(k_se3_p_noise, dim3(grid_for_tiles((n + kTile - 1) / kTile)), dim3(kBlock), 0, (hipStream_t)s, trap_row,
                     sigma, shift_scale, mean_rot, mean_shift, axes, unif, znorm, seed, rng_offset, index_base, shared_rot,
                     out_rot, out_shift, n);
  return check_launch();
}

int so3x_rigid_move(so3x_stream_t s, const float* rot, const float* shift, const float* pos, const float* frames,
                    float* out_pos, float* out_frames, int64_t S, int64_t L) {
  if (S < 0 || L <= 0 || S > 0x7fffffff || (S && (!rot || !shift || !pos || !out_pos)) || ((frames == nullptr) != (out_frames == nullptr)))
    return SO3X_ERR_INVALID_ARG;
  if (S == 0) return SO3X_OK;
  hipLaunchKernelGGL(k_rigid_move, dim3((unsigned)S), dim3(kBlock), 0, (hipStream_t)s, rot, shift, pos, frames, out_pos,
                     out_frames, L, (const int64_t*)nullptr);
  return check_launch();
}

int so3x_rigid_move_ragged(so3x_stream_t s, const float* rot, const float* shift, const float* pos, const float* frames, const int64_t* off,
                           float* out_pos, float* out_frames, int64_t S) {
  if (S < 0 || S > 0x7fffffff || (S && (!rot || !shift || !pos || !out_pos || !off)) || ((frames == nullptr) != (out_frames == nullptr)))
    return SO3X_ERR_INVALID_ARG;
  if (S == 0) return SO3X_OK;
  hipLaunchKernelGGL(k_rigid_move, dim3((unsigned)S), dim3(kBlock), 0, (hipStream_t)s, rot, shift, pos, frames, out_pos,
                     out_frames, (int64_t)0, off);
  return check_launch();
}

}  // extern "C"
