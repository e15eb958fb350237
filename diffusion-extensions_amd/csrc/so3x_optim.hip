// so3x_optim.hip -- the optimizer update of the training loop (reference so3_train.py:64,76: torch.optim.Adam(lr=3e-4),
// optim.step()) on ONE flat fp32 buffer: all 17,358 parameters of the score network (392,448 of the wide one) in a
// single launch, with the step count resident on the device so that a captured hipGraph of the training step advances
// it by itself.
#include <math.h>
#include "so3x_common.hpp"

using namespace so3x;

namespace {

// torch.optim.Adam with amsgrad=False, maximize=False (torch/optim/adam.py, _single_tensor_adam), per element:
//   g    = grad * grad_scale (+ weight_decay * p)
//   m    = m + (1 - beta1) (g - m)                      (exp_avg.lerp_(grad, 1 - beta1))
//   v    = beta2 v + (1 - beta2) g g                    (exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1 - beta2))
//   p   -= (lr / (1 - beta1^k)) * m / (sqrt(v) / sqrt(1 - beta2^k) + eps),   k = the incremented step count
// The scalars are formed in double and rounded once, as torch forms them in Python floats.
__global__ void __launch_bounds__(256)
k_adam(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, float* __restrict__ step,
       unsigned* __restrict__ ticket, int64_t n, float lr, float beta1, float beta2, float eps, float weight_decay, float grad_scale) {
  __shared__ float sc[2];
  if (threadIdx.x == 0) {
    const double k = (double)step[0] + 1.0;
    const double bc1 = 1.0 - pow((double)beta1, k), bc2 = 1.0 - pow((double)beta2, k);
    sc[0] = (float)(-(double)lr / bc1);
    sc[1] = (float)sqrt(bc2);
  }
  __syncthreads();
  const float neg_step_size = sc[0], bc2_sqrt = sc[1];
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  // A gradient whose FIRST entry is not finite is the all-NaN gradient of a training step that gave up on a hand-shake
  // (so3x_train_bwd_reduce poisons every entry, and a summed all-reduce keeps it that way on every rank): the whole update is
  // skipped, step count included -- the parameters must not be written from partial slabs.  (torch.optim.Adam would propagate
  // the NaN into the parameters; this is a deliberate deviation for that one case.)
  const bool bad = !(fabsf(g[0]) <= 3.0e38f);
  if (i < n && !bad) {
    float gi = g[i] * grad_scale;
    float pi = p[i];
    if (weight_decay != 0.0f) gi = fmaf(weight_decay, pi, gi);
    float mi = m[i], vi = v[i];
    mi = mi + (1.0f - beta1) * (gi - mi);
    vi = vi * beta2 + (1.0f - beta2) * gi * gi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] = pi + neg_step_size * (mi / denom);
    m[i] = mi;
    v[i] = vi;
  }
  // every block has read step[0] above (thread 0, before the barrier); the last block to arrive advances it
  if (threadIdx.x == 0) {
    const unsigned mine = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (mine == gridDim.x - 1) {
      __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (!bad) step[0] = step[0] + 1.0f;
    }
  }
}

}  // namespace

extern "C" {

int so3x_adam_step(so3x_stream_t s, float* params, const float* grad, float* exp_avg, float* exp_avg_sq, float* step, int64_t n,
                   float lr, float beta1, float beta2, float eps, float weight_decay, float grad_scale) {
  if (n < 0 || (n && (!params || !grad || !exp_avg || !exp_avg_sq)) || !step) return SO3X_ERR_INVALID_ARG;
  if (n == 0) return SO3X_OK;
  // step[0] = the count, step[1] = the arrival ticket of the launch (both zero-initialised by the caller, once)
  hipLaunchKernelGGL(k_adam, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)s, params, grad, exp_avg, exp_avg_sq, step,
                     reinterpret_cast<unsigned*>(step + 1), n, lr, beta1, beta2, eps, weight_decay, grad_scale);
  return check_launch();
}

}  // extern "C"
