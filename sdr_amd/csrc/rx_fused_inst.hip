// rx_fused_inst.hip -- instantiations of the fused receive kernel for ONE n_fft (SDR_LOG2N),
// compiled once per tuned size so the sizes build in parallel.
#include "common.hpp"
#include "kernels.hpp"
#include "rx_fused.hpp"

#ifndef SDR_LOG2N
#error "compile with -DSDR_LOG2N=<6..12>"
#endif

namespace sdr {

#define SDR_CAT_(a, b) a##b
#define SDR_CAT(a, b) SDR_CAT_(a, b)

// mod = 0: FFT only.  full: all outputs optional (hard bits, bins, counters); otherwise LLR only.
int SDR_CAT(rx_fused_dispatch_, SDR_LOG2N)(sdr_ctx* ctx, const RxParams& p, int mod, bool full)
{
    constexpr int L = SDR_LOG2N;
    switch (mod) {
        case 0: return launch_rx_fused<L, 0, 0, true>(ctx, p);
        case 1: return launch_rx_fused<L, 1, 0, true>(ctx, p);      // BPSK: full variant only
        case 2: return full ? launch_rx_fused<L, 1, 1, true>(ctx, p) : launch_rx_fused<L, 1, 1, false>(ctx, p);
        case 4: return full ? launch_rx_fused<L, 2, 2, true>(ctx, p) : launch_rx_fused<L, 2, 2, false>(ctx, p);
        case 6: return full ? launch_rx_fused<L, 3, 3, true>(ctx, p) : launch_rx_fused<L, 3, 3, false>(ctx, p);
        case 8: return full ? launch_rx_fused<L, 4, 4, true>(ctx, p) : launch_rx_fused<L, 4, 4, false>(ctx, p);
        default: return fail(SDR_ERR_UNSUPPORTED, "modulation %d has no fused kernel", mod);
    }
}

} // namespace sdr
