// encoder.hip — placeholder until the encoder kernels land (next commit).
#include "common.h"
using namespace icrec;
extern "C" {
size_t icrec_encoder_weight_count(const icrec_bert_cfg*) { return 0; }
int icrec_encoder_create(const float*, size_t, const icrec_bert_cfg*, int, icrec_encoder**) { set_error("encoder not built yet"); return ICREC_EINVAL; }
int icrec_encoder_destroy(icrec_encoder*) { return ICREC_OK; }
size_t icrec_encode_workspace_bytes(const icrec_encoder*, int64_t, int32_t) { return 0; }
int icrec_encode(icrec_encoder*, const int32_t*, const int32_t*, int32_t, int64_t, int32_t, float*, void*, size_t, void*) { set_error("encoder not built yet"); return ICREC_EINVAL; }
}
