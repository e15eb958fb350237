// so3x_protnet_bf16.hip -- placeholder while the exact-fp32 form is brought up (replaced by the bf16 matrix-core form)
#include "so3x_protnet.hpp"
namespace so3x { namespace prot {
bool bf16_supported(const Dims&) { return false; }
size_t bf16_workspace_bytes(const Dims&, int64_t, int64_t) { return 0; }
int forward_bf16(hipStream_t, const Dims&, const float*, const float*, const float*, const float*, const int64_t*, int64_t, const float*, const float*,
                 const float*, const int64_t*, int64_t, const int64_t*, float*, float*, void*) { return SO3X_ERR_UNSUPPORTED; }
} }
