#include "icl_common.h"
void icl_model_free(icl_ctx *ctx) {}
#define STUB(sig) extern "C" int sig { return icl_fail(nullptr, ICL_ERR_UNSUPPORTED, "not built yet"); }
STUB(icl_model_load_onnx(icl_ctx *, const char *))
STUB(icl_model_load_blob(icl_ctx *, const void *, int64_t))
STUB(icl_model_load_synthetic(icl_ctx *, uint64_t))
extern "C" int64_t icl_synthetic_blob_bytes(void) { return 0; }
STUB(icl_synthetic_blob(uint64_t, void *, int64_t))
STUB(icl_embed_u8(icl_ctx *, const uint8_t *, int64_t, int, int, float *))
STUB(icl_embed_u8_dev(icl_ctx *, const uint8_t *, int64_t, int, int, float *))
STUB(icl_embed_file(icl_ctx *, const char *, int, float *))
STUB(icl_preprocess_u8(const uint8_t *, float *))
