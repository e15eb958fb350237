// so3x_reverse_step.hpp -- one reverse-diffusion update of a unit-quaternion state, shared by the
// chain-resident samplers (so3x_diffusion.hip: 65-wide score MLP; so3x_resnet.hip: 255-wide residual net).
#pragma once
#include "so3x_math.hpp"
#include "so3x_igso3.hpp"

namespace so3x {

// rows of the [13][T] schedule table (so3x_schedule_from_betas)
enum { S_SQRT_AC = 3, S_SQRT_1MAC = 4, S_RECIP = 6, S_RECIPM1 = 7, S_COEF1 = 10, S_COEF2 = 11, S_SIGMA = 12 };

// SO3Diffusion.p_sample for one sample given the network output v (diffusion.py:291-326), in quaternion form:
//   x0hat = exp(a log x) exp(b v)^T,   mean = exp(c1 log x0hat) exp(c2 log x),   x' = mean @ IGSO3(sigma_t) (t > 0).
// idc = clamped sample index for the explicit-draw arrays, gidx = global index keying the Philox counter.
// FAST: hardware sine / cosine for the five exponentials (so3x_math.hpp sincos_sel; the bf16 chain kernels).
// the four schedule scalars of a step, for callers that fetch them ahead of time (scalar loads issued at the top of the step)
struct StepCoef { float a, b, c1, c2; };

template <bool FAST = false>
__device__ __forceinline__ Quat reverse_step(Quat q, const float (&v)[3], const float* __restrict__ sched, int T, int t,
                                             const float* __restrict__ trap_p, const uint16_t* __restrict__ guide_p,
                                             const float* __restrict__ axes,
                                             const float* __restrict__ unif, int64_t idc, uint64_t seed, uint64_t rng_offset,
                                             uint64_t gidx, const float* row_l = nullptr, const uint16_t* grow_l = nullptr,
                                             const float* knots_l = nullptr, const StepCoef* coef = nullptr) {
  const float a = coef ? coef->a : sched[S_RECIP * T + t], b = coef ? coef->b : sched[S_RECIPM1 * T + t];
  const float c1 = coef ? coef->c1 : sched[S_COEF1 * T + t], c2 = coef ? coef->c2 : sched[S_COEF2 * T + t];
  float ax[3], axh[3], vax[3];
  const float th = quat_axis_angle(q, ax);
  const float vn = fsqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
  const float vinv = vn > 0.f ? frcp(vn) : 0.f;
  vax[0] = v[0] * vinv; vax[1] = v[1] * vinv; vax[2] = v[2] * vinv;
  const Quat qh = qmul(quat_axis_angle_exp<FAST>(ax, a * th), quat_axis_angle_exp<FAST>(vax, -b * vn));
  const float thh = quat_axis_angle(qh, axh);
  q = qmul(quat_axis_angle_exp<FAST>(axh, c1 * thh), quat_axis_angle_exp<FAST>(ax, c2 * th));
  if (t != 0) {  // diffusion.py:320-326 -- no noise at t == 0
    float nax[3], u;
    if (axes) {
      float a0 = axes[idc * 3], a1 = axes[idc * 3 + 1], a2 = axes[idc * 3 + 2];
      float nrm = sqrtf(a0 * a0 + a1 * a1 + a2 * a2);
      nax[0] = a0 / nrm; nax[1] = a1 / nrm; nax[2] = a2 / nrm;
      float n2 = sqrtf(nax[0] * nax[0] + nax[1] * nax[1] + nax[2] * nax[2]);
      nax[0] /= n2; nax[1] /= n2; nax[2] /= n2;
      u = unif[idc];
    } else {
      Philox4 r = philox4x32_10(seed, gidx, rng_offset + (uint64_t)t);
      unit_axis(r.x, r.y, nax);
      u = u01(r.z);
    }
    // row_l / grow_l / knots_l: this timestep's CDF row, guide row and the knots staged in LDS by the caller (the chain kernel)
    const float* row = row_l ? row_l : trap_p + (size_t)t * 999;  // IsotropicGaussianSO3(model_stdev[0]), :325
    const uint16_t* grow = row_l ? grow_l : (guide_p ? guide_p + (size_t)t * kGuidePitch : nullptr);  // optional search guide (bit-identical)
    const float* kn = knots_l ? knots_l : SO3X_KNOTS_DATA;
    // SO3X_CHAIN_SEARCH=1 (A/B): on the LDS-staged row, the guided search as [guide] -> [8-knot window] with 9-ary narrowing of wide
    // brackets (igso3_angle_windowed: two dependent LDS round trips for 96 % of the lanes instead of the bisection's, which a wave
    // walks for its WORST lane: 6-9) -- the same index, bit-identical angles
#ifndef SO3X_CHAIN_SEARCH
#define SO3X_CHAIN_SEARCH 0
#endif
    float ang;
    if (axes) ang = igso3_angle<true>(row, row, kn, u, grow);
    else if (SO3X_CHAIN_SEARCH && row_l && grow) ang = igso3_angle_windowed<false>(row, row, kn, u, grow);
    else ang = igso3_angle<false>(row, row, kn, u, grow);
    q = qmul(q, quat_axis_angle_exp<FAST>(nax, ang));   // model_mean @ sample, :326
  }
  // no per-step renormalisation: q is rebuilt from (axis, angle) pairs every step, so its norm error is
  // three products' rounding (~3e-7) however long the chain is
  return q;
}

}  // namespace so3x
