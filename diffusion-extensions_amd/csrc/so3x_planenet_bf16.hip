// so3x_planenet_bf16.hip -- the bf16 matrix-core form of the PlaneNet denoiser (placeholder until the kernels land).
#include "so3x_planenet.hpp"

namespace so3x {
namespace plane {
bool bf16_supported(const Shape&) { return false; }
size_t bf16_workspace_bytes(const Shape&) { return 0; }
size_t bf16_stash_bytes(const Shape&) { return 0; }
int forward_bf16(hipStream_t, const Shape&, const float*, const float*, const int64_t*, float*, float*, void*, void*) { return SO3X_ERR_UNSUPPORTED; }
int backward_bf16(hipStream_t, const Shape&, const float*, const float*, const int64_t*, const float*, float*, const void*, void*) {
  return SO3X_ERR_UNSUPPORTED;
}
}  // namespace plane
}  // namespace so3x
