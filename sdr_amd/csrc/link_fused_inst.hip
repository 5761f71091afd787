// link_fused_inst.hip -- instantiations of the tuned transmit / generator / Monte-Carlo kernels
// for ONE n_fft (SDR_LOG2N); compiled once per tuned size.
#include "common.hpp"
#include "kernels.hpp"
#include "link_fused.hpp"

#ifndef SDR_LOG2N
#error "compile with -DSDR_LOG2N=<6..12>"
#endif

namespace sdr {

#define SDR_CAT_(a, b) a##b
#define SDR_CAT(a, b) SDR_CAT_(a, b)

// mod = 0: plain batched ofdm::tx from bins in memory; otherwise the Philox generator + AWGN.
int SDR_CAT(tx_fused_dispatch_, SDR_LOG2N)(sdr_ctx* ctx, const TxParams& p, int mod)
{
    constexpr int L = SDR_LOG2N;
    switch (mod) {
        case 0: return launch_tx_fused<L, 0, 0, true, false>(ctx, p);
        case 1: return launch_tx_fused<L, 1, 0, false, true>(ctx, p);
        case 2: return launch_tx_fused<L, 1, 1, false, true>(ctx, p);
        case 4: return launch_tx_fused<L, 2, 2, false, true>(ctx, p);
        case 6: return launch_tx_fused<L, 3, 3, false, true>(ctx, p);
        case 8: return launch_tx_fused<L, 4, 4, false, true>(ctx, p);
        case 10: return launch_tx_fused<L, 5, 5, false, true>(ctx, p);
        default: return fail(SDR_ERR_UNSUPPORTED, "modulation %d has no fused generator", mod);
    }
}

int SDR_CAT(mc_fused_dispatch_, SDR_LOG2N)(sdr_ctx* ctx, const McParams& p, int mod)
{
    constexpr int L = SDR_LOG2N;
    switch (mod) {
        case 1: return launch_mc_fused<L, 1, 0>(ctx, p);
        case 2: return launch_mc_fused<L, 1, 1>(ctx, p);
        case 4: return launch_mc_fused<L, 2, 2>(ctx, p);
        case 6: return launch_mc_fused<L, 3, 3>(ctx, p);
        case 8: return launch_mc_fused<L, 4, 4>(ctx, p);
        case 10: return launch_mc_fused<L, 5, 5>(ctx, p);
        default: return fail(SDR_ERR_UNSUPPORTED, "modulation %d has no fused Monte-Carlo kernel", mod);
    }
}

int SDR_CAT(chain_fused_dispatch_, SDR_LOG2N)(sdr_ctx* ctx, const ChainParams& p, int mod)
{
    constexpr int L = SDR_LOG2N;
    switch (mod) {
        case 1: return launch_chain_fused<L, 1, 0>(ctx, p);
        case 2: return launch_chain_fused<L, 1, 1>(ctx, p);
        case 4: return launch_chain_fused<L, 2, 2>(ctx, p);
        case 6: return launch_chain_fused<L, 3, 3>(ctx, p);
        case 8: return launch_chain_fused<L, 4, 4>(ctx, p);
        case 10: return launch_chain_fused<L, 5, 5>(ctx, p);
        default: return fail(SDR_ERR_UNSUPPORTED, "modulation %d has no fused link kernel", mod);
    }
}

int SDR_CAT(sweep_fused_dispatch_, SDR_LOG2N)(sdr_ctx* ctx, const SweepPointDev* d_pts, uint32_t n_points, unsigned long long blocks, const float2* tw, const float2* rng,
                                               unsigned long long seed, unsigned long long* counters)
{
    return launch_sweep_fused<SDR_LOG2N>(ctx, d_pts, n_points, blocks, tw, rng, seed, counters);
}

int SDR_CAT(sweep_symbols_per_group_, SDR_LOG2N)() { return PlanOps<SDR_LOG2N>::kSPW; }

} // namespace sdr
