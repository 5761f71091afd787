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

// mod = 0: FFT only.  variant bit 0: per-subcarrier taps / weights are given (prefetching
// instantiation); bit 1: every output optional (hard bits, bins, counters); 4: hard bits only;
// 9: LLRs only, taps without a weight array (no weight loads: scalar weight, or derived from the taps);
// 10: LLRs written and hard decisions counted, nothing else, no taps: outputs fixed at compile time (dispatched where it measured faster).
int SDR_CAT(rx_fused_dispatch_, SDR_LOG2N)(sdr_ctx* ctx, const RxParams& p, int mod, int variant)
{
    constexpr int L = SDR_LOG2N;
#define SDR_RX_CASE(MI, MQ)                                                    \
    switch (variant) {                                                         \
        case 0: return launch_rx_fused<L, MI, MQ, false, false>(ctx, p);       \
        case 1: return launch_rx_fused<L, MI, MQ, false, true>(ctx, p);        \
        case 9: return launch_rx_fused<L, MI, MQ, false, true, false, true>(ctx, p); \
        case 2: return launch_rx_fused<L, MI, MQ, true, false>(ctx, p);        \
        case 4: return launch_rx_fused<L, MI, MQ, false, false, true>(ctx, p);        \
        case 10: return launch_rx_fused<L, MI, MQ, true, false, false, false, true>(ctx, p); \
        default: return launch_rx_fused<L, MI, MQ, true, true>(ctx, p);        \
    }
    switch (mod) {
        case 0: return launch_rx_fused<L, 0, 0, true, false>(ctx, p);
        case 1:                                                       // BPSK: the LLR-only variants, or the every-output ones (no hard-bits-only form)
            if (variant == 0) return launch_rx_fused<L, 1, 0, false, false>(ctx, p);
            if (variant == 1) return launch_rx_fused<L, 1, 0, false, true>(ctx, p);
            if (variant == 9) return launch_rx_fused<L, 1, 0, false, true, false, true>(ctx, p);
            return (variant & 1) ? launch_rx_fused<L, 1, 0, true, true>(ctx, p) : launch_rx_fused<L, 1, 0, true, false>(ctx, p);
        case 2: SDR_RX_CASE(1, 1)
        case 4: SDR_RX_CASE(2, 2)
        case 6: SDR_RX_CASE(3, 3)
        case 8: SDR_RX_CASE(4, 4)
        case 10: SDR_RX_CASE(5, 5)
        default: return fail(SDR_ERR_UNSUPPORTED, "modulation %d has no fused kernel", mod);
    }
#undef SDR_RX_CASE
}

} // namespace sdr
