// p2s_assoc.hip -- multi-person association kernels (placeholder until the kernels land).
#include <hip/hip_runtime.h>
#include "p2s.h"
#include "p2s_internal.h"

extern "C" {
int p2s_associate_device(p2s_ctx *, int64_t, int32_t, int32_t, int32_t, const int32_t *, const int64_t *,
                         const void *, const p2s_assoc_params *, double *) { return P2S_ERR_INVALID_ARG; }
int p2s_associate_host(p2s_ctx *, int64_t, int32_t, int32_t, int32_t, const int32_t *, const int64_t *,
                       const void *, const p2s_assoc_params *, double *) { return P2S_ERR_INVALID_ARG; }
}
