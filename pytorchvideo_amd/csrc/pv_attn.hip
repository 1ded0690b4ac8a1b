// placeholder until the fused attention kernel lands (same ABI)
#include "pv_common.h"
extern "C" int pv_attention(const pv_attention_desc* d, pv_stream_t stream) {
  (void)d; (void)stream;
  return PV_ERR_UNSUPPORTED;
}
