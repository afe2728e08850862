// f16 Gemm (extension: the reference has no f16 kernel). Placeholder until the MFMA kernel lands.
#include "wg_internal.hpp"

int wgk_gemm_f16(wg_ctx *, bool, uint32_t, uint32_t, uint32_t, uint32_t, __half *, uint32_t, uint64_t, wgk_mat, wgk_mat) {
    return wg_set_error(WG_ERR_UNSUPPORTED, "Gemm: f16 is not implemented yet");
}
