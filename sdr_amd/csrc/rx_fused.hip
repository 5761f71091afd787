// rx_fused.hip -- placeholder until the tuned kernel lands (next commit).
#include "common.hpp"
#include "kernels.hpp"
namespace sdr {
bool rx_fused_supported(uint32_t, uint32_t) { return false; }
int rx_fused_launch(sdr_ctx*, const sdr_demod_args&) { return fail(SDR_ERR_UNSUPPORTED, "fused kernel not built"); }
int rx_fft_only_launch(sdr_ctx*, const void*, uint32_t, uint32_t, uint64_t, void*) { return fail(SDR_ERR_UNSUPPORTED, "fused kernel not built"); }
}
