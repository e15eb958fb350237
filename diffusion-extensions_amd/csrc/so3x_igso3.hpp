// so3x_igso3.hpp -- IGSO(3) closed-form density in fp64 (device) + the knot tables.
#pragma once
#include <hip/hip_runtime.h>
#include "so3x_math.hpp"

#define SO3X_KNOTS_QUAL static __device__
#include "so3x_knots.inc"
#undef SO3X_KNOTS_QUAL

namespace so3x {

// IsotropicGaussianSO3._eps_ft, distributions.py:53-72: all float64, the reference's
// operation order, inf/NaN -> 0 (:61-62; the small-eps overflow of exp(pi*t/v) is relied
// upon), and the t == 0 limit expression (:68-71, not cleaned).
__device__ __forceinline__ double eps_ft_f64(double t, double eps) {
  const double pi = 3.14159265358979323846;
  const double v = eps * eps;
  if (t == 0.0) {
    return sqrt(pi) * (v * exp(2 * pi * pi / v) - 2 * v * exp(pi * pi / v) + 4 * pi * pi * v * exp(pi * pi / v)) *
           exp(v / 4 - (2 * pi * pi) / v) / (v * v * sqrt(v));
  }
  double vals = sqrt(pi) * (1.0 / (v * sqrt(v))) * exp(v / 4) * exp(-((t / 2) * (t / 2)) / v) *
                (t - exp((-pi * pi) / v) * ((t - 2 * pi) * exp(pi * t / v) + (t + 2 * pi) * exp(-pi * t / v))) /
                (2 * sin(t / 2));
  if (isinf(vals) || isnan(vals)) vals = 0.0;
  return vals;
}

// Same density plus d log f / d omega (what autograd of distributions.py:74-77 yields
// through the fp64 island): f'/f = -t/(2v) + g'/g - cot(t/2)/2 with
// g = t - E[(t-2pi) e^{pi t/v} + (t+2pi) e^{-pi t/v}], E = e^{-pi^2/v}.
__device__ __forceinline__ double eps_ft_and_dlog_f64(double t, double eps, double* dlogf) {
  const double pi = 3.14159265358979323846;
  const double v = eps * eps;
  if (t == 0.0) { *dlogf = 0.0; return eps_ft_f64(t, eps); }
  const double E = exp((-pi * pi) / v), ep = exp(pi * t / v), em = exp(-pi * t / v);
  const double g = t - E * ((t - 2 * pi) * ep + (t + 2 * pi) * em);
  const double gp = 1 - E * (ep + (t - 2 * pi) * (pi / v) * ep + em - (t + 2 * pi) * (pi / v) * em);
  double sh, ch;
  sincos(t / 2, &sh, &ch);
  double vals = sqrt(pi) * (1.0 / (v * sqrt(v))) * exp(v / 4) * exp(-((t / 2) * (t / 2)) / v) * g / (2 * sh);
  if (isinf(vals) || isnan(vals)) vals = 0.0;
  *dlogf = -t / (2 * v) + gp / g - 0.5 * ch / sh;
  return vals;
}

// fp32 evaluation of log(float(f)) and f'/f that reproduces the fp64 island above to ~2e-6
// relative (validated against it on 4e5 random (omega, eps) pairs, eps in [0.005, 1.5]: finite /
// -inf pattern identical, max rel err 2e-5 in the fp32-denormal band).  It exists because the fp64
// form makes the kernel VALU-bound at 27 % of HBM peak; this one keeps it HBM-bound.
//   log f = 1/2 ln pi - 3/2 ln v + v/4 - w^2/(4v) + ln g - ln(2 sin(w/2)),  v = eps^2
//   g     = w - (w - 2pi) e1 - (w + 2pi) e2,   e1 = e^{-pi(pi-w)/v}, e2 = e^{-pi(pi+w)/v}
//           (combined exponents: nothing overflows; for x = pi w / v < 1 the difference e1 - e2 =
//           2 E sinh x is taken from the series so small angles do not cancel)
//   f'/f  = -w/(2v) + [4 pi E S(x) - 2 w x E sinh x]/(g w) - (cot(w/2)/2 - 1/w),  S = x cosh x - sinh x
//           (the 1/w poles of g'/g and cot/2 are removed analytically; series for w < 0.5)
// Reference behaviours reproduced explicitly:
//   * (w - 2pi) e^{pi w / v} overflows float64  <=>  x + ln(2pi - w) > ln(DBL_MAX): density zeroed -> -inf
//   * the .float() cast: values below the fp32 normal range are rounded on the 2^-149 grid.
// Caller handles w == 0 with the fp64 limit expression.
// sin_om >= 0 and cos_om are sin/cos of the FULL angle as read off the rotation matrix (log3_sc); the half-angle
// pair comes from them with one sqrt and one rcp instead of a range-reduced sincos:
//   cos(w/2) = sqrt((1+c)/2), sin(w/2) = s / (2 cos(w/2))        for c >= 0   (no cancellation in 1 + c)
//   sin(w/2) = sqrt((1-c)/2), cos(w/2) = s / (2 sin(w/2))        for c <  0   (no cancellation in 1 - c)
__device__ __forceinline__ float igso3_logf_dlog_f32(float om, float eps, float sin_om, float cos_om, float* dlogf) {
  const float PI = 3.14159274101257324f, PI_LO = -8.74227765734758577e-8f;
  const float v = eps * eps;
  const float inv_v = frcp(v);
  const float piv = PI * inv_v;
  const float x = piv * om;
  const float pmo = (PI - om) + PI_LO, ppo = (PI + om) + PI_LO;
  const float e1 = __expf(-(piv * pmo)), e2 = __expf(-(piv * ppo));
  const float x2 = x * x;
  float Esh, ES;
  const float Ech = (e1 + e2) * 0.5f;
  if (x < 1.0f) {
    const float E = __expf(-(PI * piv));
    const float sh = x * (1.0f + x2 * (1.0f / 6 + x2 * (1.0f / 120 + x2 * (1.0f / 5040 + x2 * (1.0f / 362880)))));
    const float S = x * x2 * (1.0f / 3) *
                    (1.0f + x2 * (1.0f / 10 + x2 * (1.0f / 280 + x2 * (1.0f / 15120 + x2 * (1.0f / 1330560)))));
    Esh = E * sh;
    ES = E * S;
  } else {
    Esh = (e1 - e2) * 0.5f;
    ES = x * Ech - Esh;
  }
  const float g = om - (2.0f * om * Ech - 4.0f * PI * Esh);
  const float N = 4.0f * PI * ES - 2.0f * om * x * Esh;
  const float r = N * frcp(g * om);
  const float big = fsqrt(0.5f * (1.0f + fabsf(cos_om)));   // the well-conditioned one of cos(w/2), sin(w/2)
  const float small = 0.5f * sin_om * frcp(big);
  const float sh2 = cos_om >= 0.0f ? small : big, ch2 = cos_om >= 0.0f ? big : small;
  const float o2 = om * om;
  const float c = om < 0.5f ? -om * (1.0f / 12 + o2 * (1.0f / 720 + o2 * (1.0f / 30240 + o2 * (1.0f / 1209600))))
                            : 0.5f * ch2 * frcp(sh2) - frcp(om);
  *dlogf = -om * 0.5f * inv_v + r - c;
  // one logarithm for  ln g - ln(2 sin(w/2)) - 3/2 ln v  (the quotient stays far inside the fp32 range)
  float lf = 0.572364942924700087f + 0.25f * v - 0.25f * o2 * inv_v + __logf(g * frcp(2.0f * sh2) * inv_v * frsq(v));
  if (lf < -87.33654f) {  // below the fp32 normal range: emulate the cast's rounding on the 2^-149 grid
    const float q = rintf(exp2f(lf * 1.44269504088896341f + 149.0f));
    lf = __logf(q) - 103.278929903431851f;  // log(0) = -inf when the value rounds to zero
  }
  if (x > 707.0f) {  // only here can (w - 2pi) e^{pi w / v} overflow float64 (ln(2pi - w) <= 1.84): rare branch
    if (x + __logf(2.0f * PI - om) > 709.782712893384f) lf = -INFINITY;  // the reference's float64 overflow -> 0
  }
  return lf;
}

}  // namespace so3x
