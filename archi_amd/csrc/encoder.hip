// encoder.hip -- BERT encoder forward (placeholder until the kernels land).
#include "index.h"
using namespace ak;
extern "C" int ak_encoder_create(const AkBertConfig *, const void *const *, int, ak_encoder_t *) { AK_FAIL(-7, "encoder not built"); }
extern "C" int ak_encoder_destroy(ak_encoder_t) { return 0; }
extern "C" int ak_encoder_forward(ak_encoder_t, const int32_t *, const int32_t *, int, int, int, int, float *, void *) { AK_FAIL(-7, "encoder not built"); }
