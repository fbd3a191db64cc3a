// MFMA self-attention (placeholder until the fused kernel lands: reports unsupported so
// the plan uses the generic kernel).
#include "common.h"
namespace dmme {
bool attn_mfma_supported(int, int, int, int) { return false; }
int launch_attn_mfma(int, const void*, int, int, int, void*, hipStream_t) {
    set_error("attn_mfma: not built");
    return DMME_ERR_UNSUPPORTED;
}
}  // namespace dmme
