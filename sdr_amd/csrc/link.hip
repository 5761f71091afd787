// link.hip -- device frame generator, AWGN channel and Monte-Carlo link (no reference
// counterpart; modelled on the end-to-end loop app/QFDemoWindow.cpp:201-224 with a channel
// inserted between ofdm::tx and ofdm::rx).
//
// generate:  Philox index -> to_constl (modulation.hpp:78-94) -> optional x H_k -> ofdm::tx
//            (ofdm.hpp:13-24: IFFT with the reference network, /N, CP) -> + Gaussian noise.
// ber:       generate -> fused receive path -> counters, in bounded chunks of device scratch.
#include "common.hpp"
#include "kernels.hpp"
#include "qam.hpp"
#include "rng.hpp"
#include "link_fused.hpp"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace sdr {

#define SDR_DECL_LINK(L) \
    int tx_fused_dispatch_##L(sdr_ctx*, const TxParams&, int); \
    int mc_fused_dispatch_##L(sdr_ctx*, const McParams&, int); \
    int chain_fused_dispatch_##L(sdr_ctx*, const ChainParams&, int); \
    int sweep_fused_dispatch_##L(sdr_ctx*, const SweepPointDev*, uint32_t, unsigned long long, const float2*, const float2*, unsigned long long, unsigned long long*); \
    int sweep_symbols_per_group_##L();
SDR_DECL_LINK(6) SDR_DECL_LINK(7) SDR_DECL_LINK(8) SDR_DECL_LINK(9) SDR_DECL_LINK(10) SDR_DECL_LINK(11) SDR_DECL_LINK(12) SDR_DECL_LINK(13)
#undef SDR_DECL_LINK

namespace {

int tx_dispatch(sdr_ctx* ctx, unsigned log2n, const TxParams& p, int mod)
{
    switch (log2n) {
        case 6: return tx_fused_dispatch_6(ctx, p, mod);
        case 7: return tx_fused_dispatch_7(ctx, p, mod);
        case 8: return tx_fused_dispatch_8(ctx, p, mod);
        case 9: return tx_fused_dispatch_9(ctx, p, mod);
        case 10: return tx_fused_dispatch_10(ctx, p, mod);
        case 11: return tx_fused_dispatch_11(ctx, p, mod);
        case 12: return tx_fused_dispatch_12(ctx, p, mod);
        case 13: return tx_fused_dispatch_13(ctx, p, mod);
        default: return fail(SDR_ERR_UNSUPPORTED, "n_fft = 2^%u has no fused kernel", log2n);
    }
}
int mc_dispatch(sdr_ctx* ctx, unsigned log2n, const McParams& p, int mod)
{
    switch (log2n) {
        case 6: return mc_fused_dispatch_6(ctx, p, mod);
        case 7: return mc_fused_dispatch_7(ctx, p, mod);
        case 8: return mc_fused_dispatch_8(ctx, p, mod);
        case 9: return mc_fused_dispatch_9(ctx, p, mod);
        case 10: return mc_fused_dispatch_10(ctx, p, mod);
        case 11: return mc_fused_dispatch_11(ctx, p, mod);
        case 12: return mc_fused_dispatch_12(ctx, p, mod);
        case 13: return mc_fused_dispatch_13(ctx, p, mod);
        default: return fail(SDR_ERR_UNSUPPORTED, "n_fft = 2^%u has no fused kernel", log2n);
    }
}
int chain_dispatch(sdr_ctx* ctx, unsigned log2n, const ChainParams& p, int mod)
{
    switch (log2n) {
        case 6: return chain_fused_dispatch_6(ctx, p, mod);
        case 7: return chain_fused_dispatch_7(ctx, p, mod);
        case 8: return chain_fused_dispatch_8(ctx, p, mod);
        case 9: return chain_fused_dispatch_9(ctx, p, mod);
        case 10: return chain_fused_dispatch_10(ctx, p, mod);
        case 11: return chain_fused_dispatch_11(ctx, p, mod);
        case 12: return chain_fused_dispatch_12(ctx, p, mod);
        case 13: return chain_fused_dispatch_13(ctx, p, mod);
        default: return fail(SDR_ERR_UNSUPPORTED, "n_fft = 2^%u has no fused kernel", log2n);
    }
}
int sweep_dispatch(sdr_ctx* ctx, unsigned log2n, const SweepPointDev* d_pts, uint32_t n_points, unsigned long long blocks, const float2* tw, const float2* rng,
                   unsigned long long seed, unsigned long long* counters)
{
    switch (log2n) {
        case 6: return sweep_fused_dispatch_6(ctx, d_pts, n_points, blocks, tw, rng, seed, counters);
        case 7: return sweep_fused_dispatch_7(ctx, d_pts, n_points, blocks, tw, rng, seed, counters);
        case 8: return sweep_fused_dispatch_8(ctx, d_pts, n_points, blocks, tw, rng, seed, counters);
        case 9: return sweep_fused_dispatch_9(ctx, d_pts, n_points, blocks, tw, rng, seed, counters);
        case 10: return sweep_fused_dispatch_10(ctx, d_pts, n_points, blocks, tw, rng, seed, counters);
        case 11: return sweep_fused_dispatch_11(ctx, d_pts, n_points, blocks, tw, rng, seed, counters);
        case 12: return sweep_fused_dispatch_12(ctx, d_pts, n_points, blocks, tw, rng, seed, counters);
        case 13: return sweep_fused_dispatch_13(ctx, d_pts, n_points, blocks, tw, rng, seed, counters);
        default: return fail(SDR_ERR_UNSUPPORTED, "n_fft = 2^%u has no fused kernel", log2n);
    }
}
int sweep_symbols_per_group(unsigned log2n)
{
    switch (log2n) {
        case 6: return sweep_symbols_per_group_6();
        case 7: return sweep_symbols_per_group_7();
        case 8: return sweep_symbols_per_group_8();
        case 9: return sweep_symbols_per_group_9();
        case 10: return sweep_symbols_per_group_10();
        case 11: return sweep_symbols_per_group_11();
        case 12: return sweep_symbols_per_group_12();
        case 13: return sweep_symbols_per_group_13();
        default: return 1;
    }
}
bool is_aligned(const void* p, size_t a) { return (reinterpret_cast<uintptr_t>(p) % a) == 0; }
bool generic_only() { static const bool v = getenv("SDR_GENERIC_LINK") != nullptr; return v; }   // test/tuning knob

constexpr unsigned kThreads = 256;

inline unsigned grid_for(sdr_ctx* ctx, size_t work)
{
    size_t blocks = (work + kThreads - 1) / kThreads;
    const size_t cap = size_t(ctx->num_cu) * 16;
    if (blocks > cap) blocks = cap;
    if (blocks == 0) blocks = 1;
    return static_cast<unsigned>(blocks);
}

template <int MI, int MQ>
__global__ void gen_freq_kernel(uint32_t n_fft, uint64_t seed, uint64_t g0, uint64_t n_sym, uint32_t sym_per_set, float norm,
                                const float2* __restrict__ chan, float2* __restrict__ freq, uint16_t* __restrict__ index)
{
    constexpr int BPS = MI + MQ;
    const size_t total = size_t(n_sym) * n_fft;
    const uint64_t first_set = sym_per_set ? g0 / sym_per_set : 0;
    for (size_t e = blockIdx.x * size_t(blockDim.x) + threadIdx.x; e < total; e += size_t(gridDim.x) * blockDim.x) {
        const size_t s = e / n_fft;
        const uint32_t k = static_cast<uint32_t>(e - s * n_fft);
        const uint64_t g = g0 + s;
        const unsigned idx = tx_index(seed, g, k, n_fft, BPS);
        const unsigned gi = idx >> MQ, gq = idx & ((1u << MQ) - 1u);
        float li = 0.0f, lq = 0.0f;
#pragma unroll
        for (unsigned c = 0; c < (1u << MI); ++c) li = (c == gi) ? axis_level<float, MI>(c) : li;
#pragma unroll
        for (unsigned c = 0; c < (1u << MQ); ++c) lq = (c == gq) ? axis_level<float, MQ>(c) : lq;
        float xr = li * norm, xi = lq * norm;
        if (chan) {
            const size_t set = sym_per_set ? static_cast<size_t>(g / sym_per_set - first_set) : 0;
            const float2 h = chan[set * n_fft + k];
            const float tr = xr * h.x - xi * h.y;
            const float ti = xr * h.y + xi * h.x;
            xr = tr; xi = ti;
        }
        freq[e] = make_float2(xr, xi);
        if (index) index[e] = static_cast<uint16_t>(idx);
    }
}

__global__ void add_noise_kernel(uint32_t n_fft, uint32_t cp, uint64_t seed, uint64_t g0, uint64_t n_sym, float sigma, const float2* __restrict__ rng, float2* __restrict__ rx)
{
    const uint32_t len = n_fft + cp;
    const size_t total = size_t(n_sym) * len;
    for (size_t e = blockIdx.x * size_t(blockDim.x) + threadIdx.x; e < total; e += size_t(gridDim.x) * blockDim.x) {
        const size_t s = e / len;
        const uint32_t i = static_cast<uint32_t>(e - s * len);
        const uint64_t g = g0 + s;
        const float2 nz = (i < cp) ? cp_noise(seed, g, i, cp, sigma, rng) : payload_noise(seed, g, i - cp, n_fft, sigma, rng);
        float2 v = rx[e];
        v.x = v.x + nz.x;
        v.y = v.y + nz.y;
        rx[e] = v;
    }
}

// one thread per output byte (same packing rule as demap.hip)
__global__ void pack_index_kernel(const uint16_t* __restrict__ index, size_t n_sym, int bps, size_t n_bytes, uint8_t* __restrict__ out)
{
    for (size_t b = blockIdx.x * size_t(blockDim.x) + threadIdx.x; b < n_bytes; b += size_t(gridDim.x) * blockDim.x) {
        unsigned v = 0;
        for (int i = 0; i < 8; ++i) {
            const size_t pos = b * 8 + i;
            const size_t k = pos / bps;
            const int within = static_cast<int>(pos - k * bps);
            const unsigned bit = (k < n_sym) ? ((index[k] >> (bps - 1 - within)) & 1u) : 0u;
            v = (v << 1) | bit;
        }
        out[b] = static_cast<uint8_t>(v);
    }
}

int check_link(const sdr_link_args& a)
{
    if (a.struct_size != sizeof(sdr_link_args)) return fail(SDR_ERR_INVALID, "sdr_link_args.struct_size %u != %zu", a.struct_size, sizeof(sdr_link_args));
    if (!valid_mod(static_cast<int>(a.mod))) return fail(SDR_ERR_INVALID, "unknown modulation %u", a.mod);
    if (a.n_fft == 0 || !is_pow2(a.n_fft)) return fail(SDR_ERR_NOT_POW2, "%s", kPow2Message);
    if (a.cp > a.n_fft) return fail(SDR_ERR_INVALID, "cyclic prefix %u longer than the FFT %u", a.cp, a.n_fft);
    if ((size_t(a.n_fft) * a.mod) % 8 != 0) return fail(SDR_ERR_UNSUPPORTED, "n_fft*mod must be a multiple of 8 bits");
    return SDR_OK;
}

} // namespace

int link_generate(sdr_ctx* ctx, const sdr_link_args& a, bool skip_prefix)
{
    SDR_TRY(check_link(a));
    if (a.n_sym == 0) return SDR_OK;
    if (!a.d_rx) return fail(SDR_ERR_INVALID, "null d_rx");
    const float2* rng = nullptr;
    SDR_TRY(get_rng_tables(ctx, &rng));
    if (!generic_only() && rx_fused_supported(a.n_fft, a.mod) && is_aligned(a.d_rx, 8) && is_aligned(a.d_tx_bits, 4) && is_aligned(a.d_chan, 8)) {
        const void* tw = nullptr;
        SDR_TRY(get_twiddles(ctx, a.n_fft, true, false, &tw));
        TxParams p = {};
        p.rng = rng;
        p.tw_inv = static_cast<const float2*>(tw);
        p.chan = static_cast<const float2*>(a.d_chan);
        p.out = static_cast<float2*>(a.d_rx);
        p.tx_bits = a.d_tx_bits;
        p.n_sym = a.n_sym; p.seed = a.seed; p.first_symbol = a.first_symbol;
        p.cp = a.cp; p.sym_per_set = a.sym_per_set; p.sigma = a.sigma;
        p.norm = 1.0f / std::sqrt(static_cast<float>(grid_energy(static_cast<int>(a.mod))));
        p.skip_prefix = skip_prefix ? 1u : 0u;      // (the generic path below always writes the prefix)
        return tx_dispatch(ctx, ilog2(a.n_fft), p, static_cast<int>(a.mod));
    }
    const size_t total = size_t(a.n_sym) * a.n_fft;
    void* freq = nullptr;
    void* idx = nullptr;
    SDR_TRY(get_staging(ctx, 0, total * sizeof(float2), &freq));
    if (a.d_tx_bits) SDR_TRY(get_staging(ctx, 1, total * sizeof(uint16_t), &idx));
    const int mod = static_cast<int>(a.mod);
    const float norm = 1.0f / std::sqrt(static_cast<float>(grid_energy(mod)));
    const unsigned grid = grid_for(ctx, total);
    dispatch_mod(mod, [&]<int MI, int MQ>() {
        hipLaunchKernelGGL((gen_freq_kernel<MI, MQ>), dim3(grid), dim3(kThreads), 0, ctx->stream, a.n_fft, a.seed, a.first_symbol, a.n_sym, a.sym_per_set, norm,
                           static_cast<const float2*>(a.d_chan), static_cast<float2*>(freq), static_cast<uint16_t*>(idx));
    });
    SDR_HIP(hipGetLastError());
    SDR_TRY(fft_generic_f32(ctx, freq, a.d_rx, a.n_fft, true, a.n_fft, 0, size_t(a.n_fft) + a.cp, a.cp, a.cp, a.n_sym));
    const size_t samples = size_t(a.n_sym) * (size_t(a.n_fft) + a.cp);
    hipLaunchKernelGGL(add_noise_kernel, dim3(grid_for(ctx, samples)), dim3(kThreads), 0, ctx->stream, a.n_fft, a.cp, a.seed, a.first_symbol, a.n_sym, a.sigma,
                       rng, static_cast<float2*>(a.d_rx));
    SDR_HIP(hipGetLastError());
    if (a.d_tx_bits) {
        const size_t n_bytes = total * a.mod / 8;
        hipLaunchKernelGGL(pack_index_kernel, dim3(grid_for(ctx, n_bytes)), dim3(kThreads), 0, ctx->stream, static_cast<const uint16_t*>(idx), total, mod, n_bytes, a.d_tx_bits);
        SDR_HIP(hipGetLastError());
    }
    return SDR_OK;
}

// Batched ofdm::tx for the tuned sizes (bins in device memory -> samples with cyclic prefix).
bool tx_fused_ok(size_t n_fft, size_t cp, const void* d_in, const void* d_out)
{
    return n_fft <= 8192 && rx_fused_supported(static_cast<uint32_t>(n_fft), 0) && cp <= n_fft && is_aligned(d_in, 16) && is_aligned(d_out, 8);
}
int tx_fused_launch(sdr_ctx* ctx, const void* d_in, uint32_t n_fft, uint32_t cp, uint64_t n_sym, void* d_out)
{
    const void* tw = nullptr;
    SDR_TRY(get_twiddles(ctx, n_fft, true, false, &tw));
    TxParams p = {};
    p.freq = static_cast<const float2*>(d_in);
    p.tw_inv = static_cast<const float2*>(tw);
    p.out = static_cast<float2*>(d_out);
    p.n_sym = n_sym;
    p.cp = cp;
    return tx_dispatch(ctx, ilog2(n_fft), p, 0);
}

// Monte-Carlo link: symbols are produced and consumed chunk by chunk in device scratch; only
// the eight counters are outputs.  (Tuned sizes use the fully fused kernel in rx_fused.hip.)
int link_ber(sdr_ctx* ctx, const sdr_link_args& a)
{
    SDR_TRY(check_link(a));
    if (!a.d_counters) return fail(SDR_ERR_INVALID, "null d_counters");
    if (a.n_sym == 0) return SDR_OK;
    if (!generic_only() && rx_fused_supported(a.n_fft, a.mod)) {
        const void* twf = nullptr;
        const void* twi = nullptr;
        SDR_TRY(get_twiddles(ctx, a.n_fft, false, false, &twf));
        SDR_TRY(get_twiddles(ctx, a.n_fft, true, false, &twi));
        McParams p = {};
        SDR_TRY(get_rng_tables(ctx, &p.rng));
        p.tw_fwd = static_cast<const float2*>(twf);
        p.tw_inv = static_cast<const float2*>(twi);
        p.counters = reinterpret_cast<unsigned long long*>(a.d_counters);
        p.n_sym = a.n_sym; p.seed = a.seed; p.first_symbol = a.first_symbol;
        p.sigma = a.sigma; p.w0 = a.llr_weight0;
        const float e = static_cast<float>(grid_energy(static_cast<int>(a.mod)));
        p.norm = 1.0f / std::sqrt(e);
        p.inorm = std::sqrt(e);
        return mc_dispatch(ctx, ilog2(a.n_fft), p, static_cast<int>(a.mod));
    }
    const size_t sym_bytes = (size_t(a.n_fft) + a.cp) * sizeof(float2);
    size_t chunk = (size_t(256) << 20) / sym_bytes;
    if (chunk == 0) chunk = 1;
    if (chunk > a.n_sym) chunk = a.n_sym;
    void* rx = nullptr;
    void* bits = nullptr;
    SDR_TRY(get_staging(ctx, 3, chunk * sym_bytes, &rx));
    SDR_TRY(get_staging(ctx, 4, chunk * a.n_fft * a.mod / 8, &bits));
    for (uint64_t done = 0; done < a.n_sym; done += chunk) {
        const uint64_t now = (a.n_sym - done < chunk) ? a.n_sym - done : chunk;
        sdr_link_args g = a;
        g.first_symbol = a.first_symbol + done;
        g.n_sym = now;
        g.d_chan = nullptr;
        g.d_rx = rx;
        g.d_tx_bits = static_cast<uint8_t*>(bits);
        SDR_TRY(link_generate(ctx, g));
        sdr_demod_args d = {};
        d.struct_size = sizeof(sdr_demod_args);
        d.n_fft = a.n_fft; d.cp = a.cp; d.mod = a.mod; d.n_sym = now;
        d.sym_per_set = 0; d.llr_weight0 = a.llr_weight0;
        d.d_rx = rx;
        d.d_ref_hard = static_cast<const uint8_t*>(bits);
        d.d_counters = a.d_counters;
        SDR_TRY(sdr_ofdm_demod_c32(ctx, &d));
    }
    return SDR_OK;
}

// The whole Eb/N0 sweep of one rank as ONE launch (model loop: app/QFDemoWindow.cpp:201-224, one
// iteration per symbol; here every (modulation, noise level) point's symbol range at once).  The
// host builds a table of the points, costliest modulation first so that the long workgroups are
// dispatched first, and a grid in which workgroup b serves one run of symbol groups of one point;
// counters land in row `index of the point in the caller's array` of d_counters.  Results are those
// of one sdr_link_ber_c32 call per point (same random streams: they are keyed by the global
// symbol index), without 70 launches of 50 microseconds each.
int link_sweep(sdr_ctx* ctx, const sdr_sweep_args& a)
{
    if (a.struct_size != sizeof(sdr_sweep_args)) return fail(SDR_ERR_INVALID, "sdr_sweep_args.struct_size %u != %zu", a.struct_size, sizeof(sdr_sweep_args));
    if (a.n_fft == 0 || !is_pow2(a.n_fft)) return fail(SDR_ERR_NOT_POW2, "%s", kPow2Message);
    if (a.cp > a.n_fft) return fail(SDR_ERR_INVALID, "cyclic prefix %u longer than the FFT %u", a.cp, a.n_fft);
    if (a.flags & ~static_cast<uint32_t>(SDR_SWEEP_ZERO_COUNTERS)) return fail(SDR_ERR_INVALID, "sdr_sweep_args: unknown flags 0x%x", a.flags);
    if (a.n_points == 0) return SDR_OK;
    if (!a.points) return fail(SDR_ERR_INVALID, "null points");
    if (!a.d_counters) return fail(SDR_ERR_INVALID, "null d_counters");
    if (a.flags & SDR_SWEEP_ZERO_COUNTERS) SDR_HIP(hipMemsetAsync(a.d_counters, 0, size_t(a.n_points) * 8 * sizeof(uint64_t), ctx->stream));
    bool fused = !generic_only();
    for (uint32_t i = 0; i < a.n_points; ++i) {
        const sdr_sweep_point& pt = a.points[i];
        if (!valid_mod(static_cast<int>(pt.mod))) return fail(SDR_ERR_INVALID, "point %u: unknown modulation %u", i, pt.mod);
        if ((size_t(a.n_fft) * pt.mod) % 8 != 0) return fail(SDR_ERR_UNSUPPORTED, "point %u: n_fft*mod must be a multiple of 8 bits", i);
        if (pt.flags & ~static_cast<uint32_t>(SDR_POINT_HARD_ONLY)) return fail(SDR_ERR_INVALID, "point %u: unknown flags 0x%x", i, pt.flags);
        fused = fused && rx_fused_supported(a.n_fft, pt.mod);
    }
    if (!fused) {
        // sizes without a fused kernel: one generic Monte-Carlo call per point
        for (uint32_t i = 0; i < a.n_points; ++i) {
            const sdr_sweep_point& pt = a.points[i];
            sdr_link_args l = {};
            l.struct_size = sizeof(sdr_link_args);
            l.n_fft = a.n_fft; l.cp = a.cp; l.mod = pt.mod; l.seed = a.seed;
            l.first_symbol = pt.first_symbol; l.n_sym = pt.n_sym; l.sigma = pt.sigma; l.llr_weight0 = pt.llr_weight0;
            l.d_counters = a.d_counters + size_t(i) * 8;
            if (pt.flags & SDR_POINT_HARD_ONLY) {
                // the generic engine always forms LLRs: keep what counters 5 and 6 held before the call
                // (one small device-to-device round trip through the staging arena; this is not the fast path)
                void* keep = nullptr;
                SDR_TRY(get_staging(ctx, 9, 16, &keep));
                SDR_HIP(hipMemcpyAsync(keep, l.d_counters + 5, 16, hipMemcpyDeviceToDevice, ctx->stream));
                SDR_TRY(link_ber(ctx, l));
                SDR_HIP(hipMemcpyAsync(l.d_counters + 5, keep, 16, hipMemcpyDeviceToDevice, ctx->stream));
                continue;
            }
            SDR_TRY(link_ber(ctx, l));
        }
        return SDR_OK;
    }
    const unsigned log2n = ilog2(a.n_fft);
    const unsigned long long spw = static_cast<unsigned long long>(sweep_symbols_per_group(log2n));
    std::vector<uint32_t> order;
    unsigned long long total_groups = 0;
    for (uint32_t i = 0; i < a.n_points; ++i) {
        if (a.points[i].n_sym == 0) continue;
        order.push_back(i);
        total_groups += (a.points[i].n_sym + spw - 1) / spw;
    }
    if (order.empty()) return SDR_OK;
    std::stable_sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return a.points[x].mod > a.points[y].mod; });
    // sixteen workgroups per resident slot (eight slots per CU): the tail of the launch, where slots
    // run empty, is then a few per cent of it, and the 6 atomics per wave at the end of a workgroup
    // stay rare (measured alternatives: tools/sweepbench.py)
    static const long knob = [] { const char* v = getenv("SDR_SWEEP_BLOCKS_PER_SLOT"); return v ? atol(v) : 0L; }();
    const unsigned long long target = static_cast<unsigned long long>(ctx->num_cu) * 8ull * static_cast<unsigned long long>(knob > 0 ? knob : 16);
    unsigned long long gpb = (total_groups + target - 1) / target;
    // ... but a workgroup's set-up (addresses, twiddles, six atomics at the end) wants at least ~16 Ki
    // points of work behind it, as long as every resident slot still gets a workgroup
    const unsigned long long min_gpb = (16384ull + static_cast<unsigned long long>(a.n_fft) * spw - 1) / (static_cast<unsigned long long>(a.n_fft) * spw);
    const unsigned long long slots = static_cast<unsigned long long>(ctx->num_cu) * 8ull;
    const unsigned long long fill_gpb = total_groups / slots;      // largest run that still fills the slots
    if (gpb < min_gpb) gpb = min_gpb < fill_gpb ? min_gpb : (fill_gpb > gpb ? fill_gpb : gpb);
    if (gpb < 1) gpb = 1;
    if (gpb > 0xFFFFFFFFull) gpb = 0xFFFFFFFFull;
    std::vector<SweepPointDev> table(order.size());
    unsigned long long blocks = 0;
    for (size_t j = 0; j < order.size(); ++j) {
        const sdr_sweep_point& pt = a.points[order[j]];
        SweepPointDev& d = table[j];
        d.first_symbol = pt.first_symbol;
        d.n_sym = pt.n_sym;
        d.n_groups = (pt.n_sym + spw - 1) / spw;
        d.first_block = blocks;
        d.sigma = pt.sigma;
        d.w0 = pt.llr_weight0;
        const float e = static_cast<float>(grid_energy(static_cast<int>(pt.mod)));
        d.norm = 1.0f / std::sqrt(e);
        d.inorm = std::sqrt(e);
        d.mod = pt.mod;
        d.row = order[j];
        d.groups_per_block = static_cast<uint32_t>(gpb);
        d.flags = pt.flags;
        blocks += (d.n_groups + gpb - 1) / gpb;
    }
    const void* twf = nullptr;
    SDR_TRY(get_twiddles(ctx, a.n_fft, false, false, &twf));
    // The device copy of the table is kept: a sweep that repeats the previous one (a timed repetition, the next
    // noise seed of the same grid) skips the pinned-buffer hand-over and the host-to-device copy.
    void* d_table = nullptr;
    const size_t table_bytes = table.size() * sizeof(SweepPointDev);
    if (ctx->sweep_table.size() == table_bytes && ctx->sweep_table_dev && ctx->sweep_table_dev == ctx->stage[8].ptr &&
        memcmp(ctx->sweep_table.data(), table.data(), table_bytes) == 0) {
        d_table = ctx->sweep_table_dev;
    } else {
        SDR_TRY(upload_table(ctx, table.data(), table_bytes, 8, &d_table));
        ctx->sweep_table.assign(reinterpret_cast<const unsigned char*>(table.data()), reinterpret_cast<const unsigned char*>(table.data()) + table_bytes);
        ctx->sweep_table_dev = d_table;
    }
    const float2* rng = nullptr;
    SDR_TRY(get_rng_tables(ctx, &rng));
    return sweep_dispatch(ctx, log2n, static_cast<const SweepPointDev*>(d_table), static_cast<uint32_t>(table.size()), blocks, static_cast<const float2*>(twf), rng, a.seed,
                          reinterpret_cast<unsigned long long*>(a.d_counters));
}

namespace {

__global__ void channel_kernel(uint32_t n_fft, uint64_t seed, uint64_t first_set, size_t n_sets, const float2* __restrict__ rng, float2* __restrict__ chan)
{
    const size_t total = n_sets * n_fft;
    for (size_t e = blockIdx.x * size_t(blockDim.x) + threadIdx.x; e < total; e += size_t(gridDim.x) * blockDim.x) {
        const size_t s = e / n_fft;
        chan[e] = chan_sample(seed, first_set + s, static_cast<uint32_t>(e - s * n_fft), rng);
    }
}

struct ChainLayout {
    size_t rx[2], bits[2], chan[2], taps[2], wgt[2], llr, total;
    uint64_t chunk, sets_per_chunk;
};

int chain_layout(const sdr_chain_args& a, ChainLayout& l)
{
    if (a.struct_size != sizeof(sdr_chain_args)) return fail(SDR_ERR_INVALID, "sdr_chain_args.struct_size %u != %zu", a.struct_size, sizeof(sdr_chain_args));
    if (!valid_mod(static_cast<int>(a.mod))) return fail(SDR_ERR_INVALID, "unknown modulation %u", a.mod);
    if (a.n_fft == 0 || !is_pow2(a.n_fft)) return fail(SDR_ERR_NOT_POW2, "%s", kPow2Message);
    if (!rx_fused_supported(a.n_fft, a.mod) || a.cp % 2 != 0 || a.cp > a.n_fft) return fail(SDR_ERR_UNSUPPORTED, "the chunked link covers n_fft 64..8192 with an even cp <= n_fft");
    if (a.sym_per_set == 0) return fail(SDR_ERR_INVALID, "sym_per_set must be >= 1");
    if (a.chunk_symbols == 0 || a.chunk_symbols % a.sym_per_set != 0 || a.first_symbol % a.sym_per_set != 0)
        return fail(SDR_ERR_INVALID, "chunk_symbols (%llu) and first_symbol (%llu) must be multiples of sym_per_set (%u)", static_cast<unsigned long long>(a.chunk_symbols),
                    static_cast<unsigned long long>(a.first_symbol), a.sym_per_set);
    l.chunk = a.chunk_symbols < a.n_sym ? a.chunk_symbols : ((a.n_sym + a.sym_per_set - 1) / a.sym_per_set) * a.sym_per_set;
    if (l.chunk == 0) l.chunk = a.sym_per_set;
    l.sets_per_chunk = l.chunk / a.sym_per_set;
    auto up = [](size_t v) { return (v + 255) & ~static_cast<size_t>(255); };
    size_t off = 0;
    for (int b = 0; b < 2; ++b) {
        l.rx[b] = off; off += up(l.chunk * (size_t(a.n_fft) + a.cp) * sizeof(float2));
        l.bits[b] = off; off += up(l.chunk * a.n_fft * a.mod / 8);
        l.chan[b] = off; off += up(l.sets_per_chunk * a.n_fft * sizeof(float2));
        l.taps[b] = off; off += up(l.sets_per_chunk * a.n_fft * sizeof(float2));
        l.wgt[b] = off; off += up(l.sets_per_chunk * a.n_fft * sizeof(float));
    }
    l.llr = off;
    if (!a.no_llr) off += up(l.chunk * a.n_fft * a.mod * sizeof(float));
    l.total = off;
    return SDR_OK;
}

// runs `fn` with the context enqueueing on `s`
struct StreamSwap {
    sdr_ctx* ctx; hipStream_t saved;
    StreamSwap(sdr_ctx* c, hipStream_t s) : ctx(c), saved(c->stream) { c->stream = s; }
    ~StreamSwap() { ctx->stream = saved; }
};

} // namespace

int link_channel(sdr_ctx* ctx, uint32_t n_fft, uint64_t seed, uint64_t first_set, size_t n_sets, void* d_chan)
{
    if (n_fft == 0 || n_sets == 0) return SDR_OK;
    if (!d_chan) return fail(SDR_ERR_INVALID, "null d_chan");
    const float2* rng = nullptr;
    SDR_TRY(get_rng_tables(ctx, &rng));
    hipLaunchKernelGGL(channel_kernel, dim3(grid_for(ctx, n_sets * n_fft)), dim3(kThreads), 0, ctx->stream, n_fft, seed, first_set, n_sets, rng, static_cast<float2*>(d_chan));
    SDR_HIP(hipGetLastError());
    return SDR_OK;
}

int link_chain_workspace(const sdr_chain_args& a, uint64_t* bytes)
{
    ChainLayout l;
    SDR_TRY(chain_layout(a, l));
    *bytes = l.total;
    return SDR_OK;
}

int link_chain(sdr_ctx* ctx, const sdr_chain_args& a)
{
    ChainLayout l;
    SDR_TRY(chain_layout(a, l));
    if (!a.d_counters) return fail(SDR_ERR_INVALID, "null d_counters");
    if (a.n_sym == 0) return SDR_OK;
    if (!a.d_workspace || a.workspace_bytes < l.total) return fail(SDR_ERR_INVALID, "workspace of %llu bytes, %zu needed", static_cast<unsigned long long>(a.workspace_bytes), l.total);
    if (reinterpret_cast<uintptr_t>(a.d_workspace) % 256 != 0) return fail(SDR_ERR_INVALID, "workspace must be 256-byte aligned");
    if (!ctx->aux_stream) {
        // SDR_CHAIN_PRIORITY (measurement knob): priority of the generator stream relative to the context's (-1 high, 1 low, 0 default)
        static const int prio = [] { const char* v = getenv("SDR_CHAIN_PRIORITY"); return v ? atoi(v) : 0; }();
        int lo = 0, hi = 0;
        SDR_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));          // lo = least, hi = greatest (numerically smaller)
        const int p = prio < 0 ? hi : prio > 0 ? lo : 0;
        SDR_HIP(hipStreamCreateWithPriority(&ctx->aux_stream, hipStreamNonBlocking, p));
    }
    static const bool serial = getenv("SDR_CHAIN_SERIAL") != nullptr;      // measurement knob: generator on the context's own stream
    hipStream_t gen_stream = serial ? ctx->stream : ctx->aux_stream;
    for (int b = 0; b < 2; ++b) {
        if (!ctx->chain_generated[b]) SDR_HIP(hipEventCreateWithFlags(&ctx->chain_generated[b], hipEventDisableTiming));
        if (!ctx->chain_consumed[b]) SDR_HIP(hipEventCreateWithFlags(&ctx->chain_consumed[b], hipEventDisableTiming));
    }
    // tables are created synchronously on first use: do it before the two streams start
    const void* tw = nullptr;
    SDR_TRY(get_twiddles(ctx, a.n_fft, false, false, &tw));
    SDR_TRY(get_twiddles(ctx, a.n_fft, true, false, &tw));
    { const float2* rng = nullptr; SDR_TRY(get_rng_tables(ctx, &rng)); }
    if (a.d_counters) { unsigned long long* slots = nullptr; SDR_TRY(get_counter_slots(ctx, &slots)); }
    unsigned char* ws = static_cast<unsigned char*>(a.d_workspace);
    hipStream_t main_stream = ctx->stream;
    // the generator stream starts behind whatever the caller has queued on the context's stream
    // (the workspace may still be in use there), via an event of its own
    SDR_HIP(hipEventRecord(ctx->chain_consumed[0], main_stream));
    SDR_HIP(hipStreamWaitEvent(gen_stream, ctx->chain_consumed[0], 0));
    const uint64_t n_chunks = (a.n_sym + l.chunk - 1) / l.chunk;
    // One kernel per chunk (round 3): bits -> constellation -> x channel -> IFFT -> + noise -> FFT -> x tap -> LLRs -> counters with the
    // samples in LDS only (chain_fused_kernel, link_fused.hpp); the auxiliary stream prepares the next chunk's channel, taps and
    // weights meanwhile.  The other form: generator and receive kernel with the samples in the workspace.  Same results bit for bit
    // (tests/test_gpu_chain.py runs both); the choice per shape is from tools/chain_forms.py (profiles/r3_chain_forms.txt): one kernel
    // wins at N = 2048 (+3 to +8 %, BPSK ... 256-QAM) and 4096 (+3 %), two kernels elsewhere (1 to 7 %: the one-kernel form holds 166 to
    // 313 registers).  SDR_CHAIN_FORM / sdr_diag_set_tuning("chain_form"): 1 = one kernel, 2 = two kernels, whatever the shape.
    const long form = rx_tuning().chain_form;
    const unsigned log2n = ilog2(a.n_fft);
    const bool fused = !generic_only() && (form == 1 || (form != 2 && (log2n == 11 || log2n == 12) && a.mod <= 8));
    auto generate = [&](uint64_t c) -> int {
        const int b = static_cast<int>(c & 1);
        const uint64_t done = c * l.chunk;
        const uint64_t now = a.n_sym - done < l.chunk ? a.n_sym - done : l.chunk;
        const uint64_t sets = (now + a.sym_per_set - 1) / a.sym_per_set;
        const uint64_t first_set = (a.first_symbol + done) / a.sym_per_set;
        StreamSwap swap(ctx, gen_stream);
        if (c >= 2) SDR_HIP(hipStreamWaitEvent(gen_stream, ctx->chain_consumed[b], 0));     // buffer b was read by chunk c-2
        SDR_TRY(link_channel(ctx, a.n_fft, a.seed, first_set, sets, ws + l.chan[b]));
        SDR_TRY(zf_taps_dev(ctx, static_cast<int>(a.mod), ws + l.chan[b], sets * a.n_fft, a.inv_n0, ws + l.taps[b], reinterpret_cast<float*>(ws + l.wgt[b])));
        if (!fused) {
            sdr_link_args g = {};
            g.struct_size = sizeof(sdr_link_args);
            g.n_fft = a.n_fft; g.cp = a.cp; g.mod = a.mod; g.seed = a.seed;
            g.first_symbol = a.first_symbol + done; g.n_sym = now; g.sym_per_set = a.sym_per_set; g.sigma = a.sigma;
            g.d_chan = ws + l.chan[b]; g.d_rx = ws + l.rx[b]; g.d_tx_bits = ws + l.bits[b];
            // the receive kernel drops the prefix on load (lib/inc/ofdm.hpp:41): the chain's generator neither draws its noise nor stores it
            SDR_TRY(link_generate(ctx, g, true));
        }
        SDR_HIP(hipEventRecord(ctx->chain_generated[b], gen_stream));
        return SDR_OK;
    };
    const float energy = static_cast<float>(grid_energy(static_cast<int>(a.mod)));
    auto run = [&]() -> int {
        SDR_TRY(generate(0));
        for (uint64_t c = 0; c < n_chunks; ++c) {
            const int b = static_cast<int>(c & 1);
            if (c + 1 < n_chunks) SDR_TRY(generate(c + 1));          // queued before the demodulation of chunk c: runs beside it
            const uint64_t done = c * l.chunk;
            const uint64_t now = a.n_sym - done < l.chunk ? a.n_sym - done : l.chunk;
            SDR_HIP(hipStreamWaitEvent(main_stream, ctx->chain_generated[b], 0));
            if (fused) {
                ChainParams q = {};
                const void* twf = nullptr;
                SDR_TRY(get_twiddles(ctx, a.n_fft, false, false, &twf));
                SDR_TRY(get_rng_tables(ctx, &q.rng));
                q.tw_fwd = static_cast<const float2*>(twf);
                q.chan = reinterpret_cast<const float2*>(ws + l.chan[b]);
                q.taps = reinterpret_cast<const float2*>(ws + l.taps[b]);
                q.wgt = reinterpret_cast<const float*>(ws + l.wgt[b]);
                q.llr = a.no_llr ? nullptr : reinterpret_cast<float*>(ws + l.llr);
                q.counters = reinterpret_cast<unsigned long long*>(a.d_counters);
                q.n_sym = now; q.seed = a.seed; q.first_symbol = a.first_symbol + done;
                q.sigma = a.sigma; q.norm = 1.0f / std::sqrt(energy); q.inorm = std::sqrt(energy);
                q.sym_per_set = a.sym_per_set;
                SDR_TRY(chain_dispatch(ctx, ilog2(a.n_fft), q, static_cast<int>(a.mod)));
            } else {
                sdr_demod_args d = {};
                d.struct_size = sizeof(sdr_demod_args);
                d.n_fft = a.n_fft; d.cp = a.cp; d.mod = a.mod; d.n_sym = now; d.sym_per_set = a.sym_per_set;
                d.d_rx = ws + l.rx[b];
                d.d_eq = ws + l.taps[b];
                d.d_llr_weight = reinterpret_cast<const float*>(ws + l.wgt[b]);
                d.d_llr = a.no_llr ? nullptr : reinterpret_cast<float*>(ws + l.llr);
                d.d_ref_hard = ws + l.bits[b];
                d.d_counters = a.d_counters;
                SDR_TRY(rx_fused_launch(ctx, d));
            }
            SDR_HIP(hipEventRecord(ctx->chain_consumed[b], main_stream));
        }
        return SDR_OK;
    };
    const int rc = run();
    if (rc != SDR_OK) {
        // A failure part-way leaves generator work queued on the auxiliary stream, still writing into the caller's
        // workspace, with nothing on the context's stream ordered behind it: drain it before the caller hears of the
        // failure and frees or reuses the workspace.  (The error text of `rc` is already recorded; failures here are ignored.)
        if (gen_stream != main_stream) (void)hipStreamSynchronize(gen_stream);
        return rc;
    }
    if (a.d_llr_chunk) *a.d_llr_chunk = a.no_llr ? nullptr : reinterpret_cast<float*>(ws + l.llr);
    return SDR_OK;
}

} // namespace sdr
