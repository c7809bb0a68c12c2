#include "icl_common.h"
extern "C" int icl_distance_mfma_dev(icl_ctx *ctx, const float *, int64_t, int32_t, float *, int64_t) { return icl_fail(ctx, ICL_ERR_UNSUPPORTED, "not built yet"); }
