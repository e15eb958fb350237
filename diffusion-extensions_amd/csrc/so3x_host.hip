// so3x_host.hip -- host-side entry points (no kernels): schedule buffers, constants.
// These are init-time host logic exactly as in the reference, which computes its
// schedule with numpy float64 on the host (diffusion.py:57-92).
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <hip/hip_runtime.h>
#include "../../include/so3x.h"
#define SO3X_KNOTS_QUAL static
#include "so3x_knots.inc"

extern "C" {

int so3x_abi_version(void) { return SO3X_ABI_VERSION; }

const char* so3x_error_string(int code) {
  switch (code) {
    case SO3X_OK: return "ok";
    case SO3X_ERR_INVALID_ARG: return "so3x: invalid argument (null pointer, negative size or bad stride)";
    case SO3X_ERR_WORKSPACE: return "so3x: workspace missing or too small";
    case SO3X_ERR_UNSUPPORTED: return "so3x: unsupported option";
    default: return code > 0 ? hipGetErrorString((hipError_t)code) : "so3x: unknown error";
  }
}

// diffusion.py:62-92, float64 then one rounding to fp32
int so3x_schedule_from_betas(const double* betas, int T, float* out) {
  if (!betas || !out || T <= 0) return SO3X_ERR_INVALID_ARG;
  double ac = 1.0;
  for (int i = 0; i < T; i++) {
    const double alpha = 1.0 - betas[i];
    const double ac_prev = (i == 0) ? 1.0 : ac;
    ac = (i == 0) ? alpha : ac * alpha;
    const double pv = betas[i] * (1. - ac_prev) / (1. - ac);
    const float logvar = (float)log(pv > 1e-20 ? pv : 1e-20);
    out[0 * T + i] = (float)betas[i];
    out[1 * T + i] = (float)ac;
    out[2 * T + i] = (float)ac_prev;
    out[3 * T + i] = (float)sqrt(ac);
    out[4 * T + i] = (float)sqrt(1. - ac);
    out[5 * T + i] = (float)log(1. - ac);
    out[6 * T + i] = (float)sqrt(1. / ac);
    out[7 * T + i] = (float)sqrt(1. / ac - 1);
    out[8 * T + i] = (float)pv;
    out[9 * T + i] = logvar;
    out[10 * T + i] = (float)(betas[i] * sqrt(ac_prev) / (1. - ac));
    out[11 * T + i] = (float)((1. - ac_prev) * sqrt(alpha) / (1. - ac));
    out[12 * T + i] = expf(0.5f * logvar);  // model_stdev, diffusion.py:324 (fp32)
  }
  return SO3X_OK;
}

// published lucidrains definition; the reference's fork is un-vendored (parity unpinned)
int so3x_cosine_beta_schedule(int T, double* betas) {
  if (!betas || T <= 0) return SO3X_ERR_INVALID_ARG;
  const int steps = T + 1;
  const double s = 0.008, pi = 3.14159265358979323846;
  double* ac = (double*)malloc(sizeof(double) * steps);
  if (!ac) return SO3X_ERR_INVALID_ARG;
  for (int i = 0; i < steps; i++) {
    const double x = (i == steps - 1) ? (double)steps : (double)i * ((double)steps / (double)(steps - 1));
    const double c = cos(((x / steps) + s) / (1 + s) * pi * 0.5);
    ac[i] = c * c;
  }
  const double a0 = ac[0];
  for (int i = 0; i < steps; i++) ac[i] /= a0;
  for (int i = 0; i < T; i++) {
    const double b = 1 - ac[i + 1] / ac[i];
    betas[i] = b < 0 ? 0 : (b > 0.999 ? 0.999 : b);
  }
  free(ac);
  return SO3X_OK;
}

int so3x_igso3_knots(float* knots, float* haar_w) {
  if (knots) memcpy(knots, SO3X_KNOTS_DATA, sizeof(SO3X_KNOTS_DATA));
  if (haar_w) memcpy(haar_w, SO3X_HAAR_W_DATA, sizeof(SO3X_HAAR_W_DATA));
  return SO3X_OK;
}

int so3x_posemb_freqs(int half_dim, float* out) {
  if (!out || half_dim < 2) return SO3X_ERR_INVALID_ARG;
  const double emb = log(10000.0) / (half_dim - 1);
  for (int i = 0; i < half_dim; i++) out[i] = expf((float)i * (float)(-emb));
  return SO3X_OK;
}

}  // extern "C"
