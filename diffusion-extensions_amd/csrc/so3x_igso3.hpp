// so3x_igso3.hpp -- IGSO(3) closed-form density in fp64 (device) + the knot tables.
#pragma once
#include <hip/hip_runtime.h>

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

}  // namespace so3x
