// capi.hip -- the extern "C" surface declared in include/sdr_amd.h: argument checks, choice
// between the tuned fused kernels and the generic engines, host-memory convenience forms.
#include "common.hpp"
#include "sdr_amd_diag.h"
#include "kernels.hpp"
#include "qam.hpp"
#include "rng.hpp"

#include <cmath>
#include <cstddef>
#include <cstring>

using namespace sdr;

namespace {

int fft2_device(sdr_ctx* ctx, bool is_double, void* d_data, size_t n, size_t batch, int inverse)
{
    SDR_TRY(ensure_device(ctx));
    if (n <= 1 || batch == 0) return SDR_OK;                       // lib/inc/fft.hpp:91: sizes 0 and 1 pass untouched
    if (!is_pow2(n)) return fail(SDR_ERR_NOT_POW2, "%s", kPow2Message);
    if (!d_data) return fail(SDR_ERR_INVALID, "null device pointer");
    // fp32 batches of a planned size run in place through the tuned kernels (a symbol is wholly in
    // registers before any of it is written back): forward = receive kernel without prefix and
    // demapper, inverse = transmit kernel without prefix
    if (!is_double && rx_fused_supported(static_cast<uint32_t>(n <= 8192 ? n : 0), 0) && reinterpret_cast<uintptr_t>(d_data) % 16 == 0) {
        if (!inverse) return rx_fft_only_launch(ctx, d_data, static_cast<uint32_t>(n), 0, batch, d_data);
        if (tx_fused_ok(n, 0, d_data, d_data)) return tx_fused_launch(ctx, d_data, static_cast<uint32_t>(n), 0, batch, d_data);
    }
    return is_double ? fft_generic_f64(ctx, d_data, d_data, n, inverse != 0, n, 0, n, 0, 0, batch)
                     : fft_generic_f32(ctx, d_data, d_data, n, inverse != 0, n, 0, n, 0, 0, batch);
}

int ofdm_rx_device(sdr_ctx* ctx, bool is_double, const void* d_in, size_t n_fft, size_t cp, size_t n_sym, void* d_out)
{
    SDR_TRY(ensure_device(ctx));
    if (n_sym == 0 || n_fft == 0) return SDR_OK;
    if (!is_pow2(n_fft)) return fail(SDR_ERR_NOT_POW2, "%s", kPow2Message);
    if (!d_in || !d_out) return fail(SDR_ERR_INVALID, "null device pointer");
    if (!is_double && n_fft <= 8192 && cp <= 0xFFFFFFFFu && cp % 2 == 0 && rx_fused_supported(static_cast<uint32_t>(n_fft), 0) &&
        reinterpret_cast<uintptr_t>(d_in) % 16 == 0 && reinterpret_cast<uintptr_t>(d_out) % 16 == 0)
        return rx_fft_only_launch(ctx, d_in, static_cast<uint32_t>(n_fft), static_cast<uint32_t>(cp), n_sym, d_out);
    return is_double ? fft_generic_f64(ctx, d_in, d_out, n_fft, false, n_fft + cp, cp, n_fft, 0, 0, n_sym)
                     : fft_generic_f32(ctx, d_in, d_out, n_fft, false, n_fft + cp, cp, n_fft, 0, 0, n_sym);
}

int ofdm_tx_device(sdr_ctx* ctx, bool is_double, const void* d_in, size_t n_fft, size_t cp, size_t n_sym, void* d_out)
{
    SDR_TRY(ensure_device(ctx));
    if (n_sym == 0) return SDR_OK;
    if (n_fft != 0 && !is_pow2(n_fft)) return fail(SDR_ERR_NOT_POW2, "%s", kPow2Message);
    if (cp > n_fft) return fail(SDR_ERR_INVALID, "batched tx needs cp (%zu) <= n_fft (%zu)", cp, n_fft);
    if (n_fft == 0) return SDR_OK;
    if (!d_in || !d_out) return fail(SDR_ERR_INVALID, "null device pointer");
    if (!is_double && tx_fused_ok(n_fft, cp, d_in, d_out))
        return tx_fused_launch(ctx, d_in, static_cast<uint32_t>(n_fft), static_cast<uint32_t>(cp), n_sym, d_out);
    return is_double ? fft_generic_f64(ctx, d_in, d_out, n_fft, true, n_fft, 0, n_fft + cp, cp, cp, n_sym)
                     : fft_generic_f32(ctx, d_in, d_out, n_fft, true, n_fft, 0, n_fft + cp, cp, cp, n_sym);
}

} // namespace

namespace sdr {

int read_demod_args(const sdr_demod_args* in, sdr_demod_args& out)
{
    if (!in) return fail(SDR_ERR_INVALID, "null args");
    constexpr size_t kLegacy = offsetof(sdr_demod_args, flags);      // the round-2 layout ends where `flags` begins
    out = sdr_demod_args{};
    if (in->struct_size == sizeof(sdr_demod_args)) out = *in;
    else if (in->struct_size == kLegacy) memcpy(&out, in, kLegacy);
    else return fail(SDR_ERR_INVALID, "sdr_demod_args.struct_size %u != %zu", in->struct_size, sizeof(sdr_demod_args));
    out.struct_size = sizeof(sdr_demod_args);
    if (out.flags & ~static_cast<uint32_t>(SDR_DEMOD_WEIGHT_FROM_TAPS)) return fail(SDR_ERR_INVALID, "sdr_demod_args: unknown flags 0x%x", out.flags);
    if ((out.flags & SDR_DEMOD_WEIGHT_FROM_TAPS) && (!out.d_eq || out.d_llr_weight))
        return fail(SDR_ERR_INVALID, "SDR_DEMOD_WEIGHT_FROM_TAPS needs d_eq and no d_llr_weight");
    if (!valid_mod(static_cast<int>(out.mod))) return fail(SDR_ERR_INVALID, "unknown modulation %u", out.mod);
    return SDR_OK;
}

} // namespace sdr

namespace {

// copy in -> run -> copy out -> synchronise
template <typename Fn>
int with_host_buffers(sdr_ctx* ctx, const void* h_in, size_t in_bytes, void* h_out, size_t out_bytes, bool in_place, Fn&& fn)
{
    SDR_TRY(ensure_device(ctx));
    void* d_in = nullptr;
    void* d_out = nullptr;
    if (in_bytes) SDR_TRY(get_staging(ctx, 6, in_bytes, &d_in));
    if (in_place) d_out = d_in;
    else if (out_bytes) SDR_TRY(get_staging(ctx, 7, out_bytes, &d_out));
    if (in_bytes) SDR_HIP(hipMemcpyAsync(d_in, h_in, in_bytes, hipMemcpyHostToDevice, ctx->stream));
    SDR_TRY(fn(d_in, d_out));
    if (out_bytes) SDR_HIP(hipMemcpyAsync(h_out, d_out, out_bytes, hipMemcpyDeviceToHost, ctx->stream));
    SDR_HIP(hipStreamSynchronize(ctx->stream));
    return SDR_OK;
}

} // namespace

extern "C" {

int sdr_fft2_c32(sdr_ctx* ctx, void* d_data, size_t n, size_t batch, int inverse) { return fft2_device(ctx, false, d_data, n, batch, inverse); }
int sdr_fft2_c64(sdr_ctx* ctx, void* d_data, size_t n, size_t batch, int inverse) { return fft2_device(ctx, true, d_data, n, batch, inverse); }

int sdr_fft2_host_c32(sdr_ctx* ctx, void* h_data, size_t n, size_t batch, int inverse)
{
    if (n <= 1 || batch == 0) return SDR_OK;
    if (!is_pow2(n)) return fail(SDR_ERR_NOT_POW2, "%s", kPow2Message);
    if (!h_data) return fail(SDR_ERR_INVALID, "null host pointer");
    const size_t bytes = n * batch * 2 * sizeof(float);
    return with_host_buffers(ctx, h_data, bytes, h_data, bytes, true, [&](void* d_in, void*) { return fft2_device(ctx, false, d_in, n, batch, inverse); });
}
int sdr_fft2_host_c64(sdr_ctx* ctx, void* h_data, size_t n, size_t batch, int inverse)
{
    if (n <= 1 || batch == 0) return SDR_OK;
    if (!is_pow2(n)) return fail(SDR_ERR_NOT_POW2, "%s", kPow2Message);
    if (!h_data) return fail(SDR_ERR_INVALID, "null host pointer");
    const size_t bytes = n * batch * 2 * sizeof(double);
    return with_host_buffers(ctx, h_data, bytes, h_data, bytes, true, [&](void* d_in, void*) { return fft2_device(ctx, true, d_in, n, batch, inverse); });
}

int sdr_ofdm_rx_c32(sdr_ctx* ctx, const void* d_in, size_t n_fft, size_t cp, size_t n_sym, void* d_out) { return ofdm_rx_device(ctx, false, d_in, n_fft, cp, n_sym, d_out); }
int sdr_ofdm_rx_c64(sdr_ctx* ctx, const void* d_in, size_t n_fft, size_t cp, size_t n_sym, void* d_out) { return ofdm_rx_device(ctx, true, d_in, n_fft, cp, n_sym, d_out); }
int sdr_ofdm_tx_c32(sdr_ctx* ctx, const void* d_in, size_t n_fft, size_t cp, size_t n_sym, void* d_out) { return ofdm_tx_device(ctx, false, d_in, n_fft, cp, n_sym, d_out); }
int sdr_ofdm_tx_c64(sdr_ctx* ctx, const void* d_in, size_t n_fft, size_t cp, size_t n_sym, void* d_out) { return ofdm_tx_device(ctx, true, d_in, n_fft, cp, n_sym, d_out); }

#define SDR_OFDM_HOST(NAME, DEVFN, IS_DOUBLE, ELEM, IN_LEN, OUT_LEN)                                                      \
    int NAME(sdr_ctx* ctx, const void* h_in, size_t n_fft, size_t cp, size_t n_sym, void* h_out)                            \
    {                                                                                                                    \
        if (n_sym == 0) return SDR_OK;                                                                                   \
        if (n_fft != 0 && !is_pow2(n_fft)) return fail(SDR_ERR_NOT_POW2, "%s", kPow2Message);                             \
        if (!h_in || !h_out) return fail(SDR_ERR_INVALID, "null host pointer");                                           \
        const size_t in_bytes = n_sym * (IN_LEN) * 2 * sizeof(ELEM);                                                       \
        const size_t out_bytes = n_sym * (OUT_LEN) * 2 * sizeof(ELEM);                                                     \
        return with_host_buffers(ctx, h_in, in_bytes, h_out, out_bytes, false,                                             \
                                 [&](void* d_in, void* d_out) { return DEVFN(ctx, IS_DOUBLE, d_in, n_fft, cp, n_sym, d_out); }); \
    }
SDR_OFDM_HOST(sdr_ofdm_rx_host_c32, ofdm_rx_device, false, float, n_fft + cp, n_fft)
SDR_OFDM_HOST(sdr_ofdm_rx_host_c64, ofdm_rx_device, true, double, n_fft + cp, n_fft)
SDR_OFDM_HOST(sdr_ofdm_tx_host_c32, ofdm_tx_device, false, float, n_fft, n_fft + cp)
SDR_OFDM_HOST(sdr_ofdm_tx_host_c64, ofdm_tx_device, true, double, n_fft, n_fft + cp)
#undef SDR_OFDM_HOST

int sdr_to_constl_c32(sdr_ctx* ctx, int mod, const uint8_t* d_bytes, size_t n_sym, void* d_out)
{
    SDR_TRY(ensure_device(ctx));
    return to_constl_dev(ctx, false, mod, d_bytes, n_sym, d_out);
}
int sdr_to_constl_c64(sdr_ctx* ctx, int mod, const uint8_t* d_bytes, size_t n_sym, void* d_out)
{
    SDR_TRY(ensure_device(ctx));
    return to_constl_dev(ctx, true, mod, d_bytes, n_sym, d_out);
}
int sdr_from_constl_c32(sdr_ctx* ctx, int mod, const void* d_sym, size_t n_sym, uint8_t* d_bytes)
{
    SDR_TRY(ensure_device(ctx));
    return from_constl_dev(ctx, false, mod, d_sym, n_sym, d_bytes);
}
int sdr_from_constl_c64(sdr_ctx* ctx, int mod, const void* d_sym, size_t n_sym, uint8_t* d_bytes)
{
    SDR_TRY(ensure_device(ctx));
    return from_constl_dev(ctx, true, mod, d_sym, n_sym, d_bytes);
}

static int to_constl_host(sdr_ctx* ctx, bool is_double, int mod, const uint8_t* h_bytes, size_t n_sym, void* h_out)
{
    if (!valid_mod(mod)) return fail(SDR_ERR_INVALID, "unknown modulation %d", mod);
    if (n_sym == 0) return SDR_OK;
    if (!h_bytes || !h_out) return fail(SDR_ERR_INVALID, "null host pointer");
    const size_t in_bytes = (n_sym * mod + 7) / 8;
    const size_t out_bytes = n_sym * 2 * (is_double ? sizeof(double) : sizeof(float));
    return with_host_buffers(ctx, h_bytes, in_bytes, h_out, out_bytes, false,
                             [&](void* d_in, void* d_out) { return to_constl_dev(ctx, is_double, mod, static_cast<const uint8_t*>(d_in), n_sym, d_out); });
}
static int from_constl_host(sdr_ctx* ctx, bool is_double, int mod, const void* h_sym, size_t n_sym, uint8_t* h_bytes)
{
    if (!valid_mod(mod)) return fail(SDR_ERR_INVALID, "unknown modulation %d", mod);
    const size_t out_bytes = (n_sym * mod) / 8;
    if (out_bytes == 0) return SDR_OK;
    if (!h_sym || !h_bytes) return fail(SDR_ERR_INVALID, "null host pointer");
    const size_t in_bytes = n_sym * 2 * (is_double ? sizeof(double) : sizeof(float));
    return with_host_buffers(ctx, h_sym, in_bytes, h_bytes, out_bytes, false,
                             [&](void* d_in, void* d_out) { return from_constl_dev(ctx, is_double, mod, d_in, n_sym, static_cast<uint8_t*>(d_out)); });
}
int sdr_to_constl_host_c32(sdr_ctx* ctx, int mod, const uint8_t* h_bytes, size_t n_sym, void* h_out) { return to_constl_host(ctx, false, mod, h_bytes, n_sym, h_out); }
int sdr_to_constl_host_c64(sdr_ctx* ctx, int mod, const uint8_t* h_bytes, size_t n_sym, void* h_out) { return to_constl_host(ctx, true, mod, h_bytes, n_sym, h_out); }
int sdr_from_constl_host_c32(sdr_ctx* ctx, int mod, const void* h_sym, size_t n_sym, uint8_t* h_bytes) { return from_constl_host(ctx, false, mod, h_sym, n_sym, h_bytes); }
int sdr_from_constl_host_c64(sdr_ctx* ctx, int mod, const void* h_sym, size_t n_sym, uint8_t* h_bytes) { return from_constl_host(ctx, true, mod, h_sym, n_sym, h_bytes); }

int sdr_nearest_c32(sdr_ctx* ctx, int mod, const void* d_sym, size_t n_sym, uint16_t* d_index)
{
    SDR_TRY(ensure_device(ctx));
    return nearest_dev(ctx, false, mod, d_sym, n_sym, d_index);
}
int sdr_nearest_c64(sdr_ctx* ctx, int mod, const void* d_sym, size_t n_sym, uint16_t* d_index)
{
    SDR_TRY(ensure_device(ctx));
    return nearest_dev(ctx, true, mod, d_sym, n_sym, d_index);
}
static int nearest_host(sdr_ctx* ctx, bool is_double, int mod, const void* h_sym, size_t n_sym, uint16_t* h_index)
{
    if (!valid_mod(mod)) return fail(SDR_ERR_INVALID, "unknown modulation %d", mod);
    if (n_sym == 0) return SDR_OK;
    if (!h_sym || !h_index) return fail(SDR_ERR_INVALID, "null host pointer");
    const size_t in_bytes = n_sym * 2 * (is_double ? sizeof(double) : sizeof(float));
    return with_host_buffers(ctx, h_sym, in_bytes, h_index, n_sym * sizeof(uint16_t), false,
                             [&](void* d_in, void* d_out) { return nearest_dev(ctx, is_double, mod, d_in, n_sym, static_cast<uint16_t*>(d_out)); });
}
int sdr_nearest_host_c32(sdr_ctx* ctx, int mod, const void* h_sym, size_t n_sym, uint16_t* h_index) { return nearest_host(ctx, false, mod, h_sym, n_sym, h_index); }
int sdr_nearest_host_c64(sdr_ctx* ctx, int mod, const void* h_sym, size_t n_sym, uint16_t* h_index) { return nearest_host(ctx, true, mod, h_sym, n_sym, h_index); }

int sdr_constellation_f32(int mod, float* norm, float* inorm, float* table)
{
    if (!valid_mod(mod)) return fail(SDR_ERR_INVALID, "unknown modulation %d", mod);
    if (!norm || !inorm || !table) return fail(SDR_ERR_INVALID, "null output");
    const float e = static_cast<float>(grid_energy(mod));
    *norm = 1.0f / std::sqrt(e);
    *inorm = std::sqrt(e);
    const int mi = mod == 1 ? 1 : mod / 2, mq = mod == 1 ? 0 : mod / 2;
    for (unsigned i = 0; i < (1u << mod); ++i) {
        const unsigned gi = i >> mq, gq = i & ((1u << mq) - 1u);
        table[2 * i] = static_cast<float>(2 * static_cast<int>(gray_to_binary(gi)) - ((1 << mi) - 1));
        table[2 * i + 1] = mq ? static_cast<float>(2 * static_cast<int>(gray_to_binary(gq)) - ((1 << mq) - 1)) : 0.0f;
    }
    return SDR_OK;
}

int sdr_ofdm_demod_c32(sdr_ctx* ctx, const sdr_demod_args* args)
{
    SDR_TRY(ensure_device(ctx));
    sdr_demod_args a;
    SDR_TRY(read_demod_args(args, a));
    if (a.n_sym == 0 || a.n_fft == 0) return SDR_OK;
    if (!is_pow2(a.n_fft)) return fail(SDR_ERR_NOT_POW2, "%s", kPow2Message);
    if (!a.d_rx) return fail(SDR_ERR_INVALID, "null d_rx");
    if ((a.d_ref_hard != nullptr) != (a.d_counters != nullptr)) return fail(SDR_ERR_INVALID, "d_ref_hard and d_counters must be given together");
    if (rx_fused_supported(a.n_fft, a.mod) && rx_fused_layout_ok(a)) return rx_fused_launch(ctx, a);
    // generic sizes: reference-network FFT into memory, then the element-wise back end
    void* d_fft = a.d_fft;
    if (!d_fft) SDR_TRY(get_staging(ctx, 5, size_t(a.n_sym) * a.n_fft * sizeof(float2), &d_fft));
    SDR_TRY(fft_generic_f32(ctx, a.d_rx, d_fft, a.n_fft, false, size_t(a.n_fft) + a.cp, a.cp, a.n_fft, 0, 0, a.n_sym));
    return demap_after_fft(ctx, a, d_fft);
}

// Placement search (include/sdr_amd.h): the traffic model is copy_shape_launch, the receive kernel's reads and writes
// in its workgroup order without the arithmetic, timed with events on the context's stream.
int sdr_placement_search_c32(sdr_ctx* ctx, void* d_arena, uint64_t arena_bytes, uint32_t n_fft, uint32_t cp, uint32_t mod, uint64_t n_sym,
                             uint64_t step_bytes, uint64_t* in_offset, uint64_t* out_offset, float* best_gbs, float* worst_gbs)
{
    SDR_TRY(ensure_device(ctx));
    if (!d_arena || !in_offset || !out_offset) return fail(SDR_ERR_INVALID, "null pointer");
    if (!valid_mod(static_cast<int>(mod)) || n_fft < 64 || n_fft > 8192 || (n_fft & (n_fft - 1)) != 0 || cp % 2 != 0 || n_sym == 0)
        return fail(SDR_ERR_INVALID, "placement search: n_fft 64 ... 8192, even cp, a known modulation, n_sym > 0");
    if (reinterpret_cast<uintptr_t>(d_arena) % 16 != 0) return fail(SDR_ERR_INVALID, "placement search: arena must be 16-byte aligned");
    constexpr uint64_t kGrain = 2ull << 20;
    auto up = [](uint64_t v) { return (v + kGrain - 1) / kGrain * kGrain; };
    const uint64_t in_bytes = up(n_sym * (static_cast<uint64_t>(n_fft) + cp) * 8);
    const uint64_t out_bytes = up(n_sym * static_cast<uint64_t>(n_fft) * mod * 4);
    if (in_bytes + out_bytes > arena_bytes) return fail(SDR_ERR_INVALID, "placement search: arena of %llu bytes, %llu needed", static_cast<unsigned long long>(arena_bytes),
                                                       static_cast<unsigned long long>(in_bytes + out_bytes));
    // default step: half the output buffer, so that some candidate straddles every boundary between two regions of the arena whatever
    // the buffer's size; at most 8 GiB, and at least 1/256 of the arena (a bound on the number of launches)
    uint64_t step = step_bytes ? up(step_bytes) : up(out_bytes / 2);
    if (!step_bytes) {
        if (step > (8ull << 30)) step = 8ull << 30;
        if (step < up(arena_bytes / 256)) step = up(arena_bytes / 256);
    }
    unsigned char* base = static_cast<unsigned char*>(d_arena);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    SDR_HIP(hipEventCreate(&e0));
    hipError_t ce = hipEventCreate(&e1);
    if (ce != hipSuccess) { (void)hipEventDestroy(e0); return fail_hip(ce, "hipEventCreate"); }
    const double alg_bytes = static_cast<double>(n_sym) * (8.0 * n_fft + 4.0 * n_fft * mod);
    int rc = SDR_OK;
    auto rate = [&](uint64_t in_off, uint64_t out_off, double* gbs) -> int {
        SDR_TRY(copy_shape_launch(ctx, base + in_off, n_fft, cp, mod, n_sym, reinterpret_cast<float*>(base + out_off)));      // first touch / warm-up
        SDR_HIP(hipEventRecord(e0, ctx->stream));
        for (int i = 0; i < 2; ++i) SDR_TRY(copy_shape_launch(ctx, base + in_off, n_fft, cp, mod, n_sym, reinterpret_cast<float*>(base + out_off)));
        SDR_HIP(hipEventRecord(e1, ctx->stream));
        SDR_HIP(hipEventSynchronize(e1));
        float ms = 0.0f;
        SDR_HIP(hipEventElapsedTime(&ms, e0, e1));
        *gbs = ms > 0.0f ? alg_bytes / (ms / 2.0 * 1e-3) / 1e9 : 0.0;
        return SDR_OK;
    };
    auto disjoint = [&](uint64_t in_off, uint64_t out_off) { return in_off + in_bytes <= out_off || out_off + out_bytes <= in_off; };
    double best = -1.0, worst = 1e30;
    uint64_t best_in = 0, best_out = 0;
    // 1. the output at every multiple of the step, the input out of its way (in front of it if it fits there, else behind it)
    for (uint64_t out_off = 0; rc == SDR_OK && out_off + out_bytes <= arena_bytes; out_off += step) {
        uint64_t in_off = 0;
        if (out_off < in_bytes) in_off = out_off + out_bytes;
        if (in_off + in_bytes > arena_bytes) continue;
        double g = 0.0;
        rc = rate(in_off, out_off, &g);
        if (rc != SDR_OK) break;
        if (g > best) { best = g; best_in = in_off; best_out = out_off; }
        if (g < worst) worst = g;
    }
    // 2. the input at every multiple of the step beside the best output
    for (uint64_t in_off = 0; rc == SDR_OK && best >= 0.0 && in_off + in_bytes <= arena_bytes; in_off += step) {
        if (!disjoint(in_off, best_out) || in_off == best_in) continue;
        double g = 0.0;
        rc = rate(in_off, best_out, &g);
        if (rc != SDR_OK) break;
        if (g > best) { best = g; best_in = in_off; }
        if (g < worst) worst = g;
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (rc != SDR_OK) return rc;
    if (best < 0.0) return fail(SDR_ERR_INVALID, "placement search: no candidate fits the arena");
    *in_offset = best_in;
    *out_offset = best_out;
    if (best_gbs) *best_gbs = static_cast<float>(best);
    if (worst_gbs) *worst_gbs = static_cast<float>(worst);
    return SDR_OK;
}

int sdr_diag_copy_shape_c32(sdr_ctx* ctx, const void* d_rx, uint32_t n_fft, uint32_t cp, uint32_t mod, uint64_t n_sym, float* d_llr)
{
    SDR_TRY(ensure_device(ctx));
    return copy_shape_launch(ctx, d_rx, n_fft, cp, mod, n_sym, d_llr);
}

int sdr_diag_set_tuning(const char* name, long value) { return rx_set_tuning(name, value); }

int sdr_zf_taps_c32(sdr_ctx* ctx, int mod, const void* d_chan, size_t n, float inv_n0, void* d_taps, float* d_weight)
{
    SDR_TRY(ensure_device(ctx));
    return zf_taps_dev(ctx, mod, d_chan, n, inv_n0, d_taps, d_weight);
}

int sdr_zf_taps_host_c32(sdr_ctx* ctx, int mod, const void* h_chan, size_t n, float inv_n0, void* h_taps, float* h_weight)
{
    SDR_TRY(ensure_device(ctx));
    if (!valid_mod(mod)) return fail(SDR_ERR_INVALID, "unknown modulation %d", mod);
    if (n == 0) return SDR_OK;
    if (!h_chan || !h_taps || !h_weight) return fail(SDR_ERR_INVALID, "null host pointer");
    auto up = [](size_t v) { return (v + 255) & ~static_cast<size_t>(255); };
    const size_t o_taps = up(n * 8), o_w = o_taps + up(n * 8);
    void* arena_v = nullptr;
    SDR_TRY(get_staging(ctx, 3, o_w + up(n * 4), &arena_v));
    unsigned char* arena = static_cast<unsigned char*>(arena_v);
    SDR_HIP(hipMemcpyAsync(arena, h_chan, n * 8, hipMemcpyHostToDevice, ctx->stream));
    SDR_TRY(zf_taps_dev(ctx, mod, arena, n, inv_n0, arena + o_taps, reinterpret_cast<float*>(arena + o_w)));
    SDR_HIP(hipMemcpyAsync(h_taps, arena + o_taps, n * 8, hipMemcpyDeviceToHost, ctx->stream));
    SDR_HIP(hipMemcpyAsync(h_weight, arena + o_w, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    SDR_HIP(hipStreamSynchronize(ctx->stream));
    return SDR_OK;
}

int sdr_soft_demap_c32(sdr_ctx* ctx, int mod, const void* d_sym, size_t n_sym, const float* d_weight, float weight0, float* d_llr)
{
    SDR_TRY(ensure_device(ctx));
    return soft_demap_dev(ctx, mod, d_sym, n_sym, d_weight, weight0, d_llr);
}

int sdr_soft_demap_host_c32(sdr_ctx* ctx, int mod, const void* h_sym, size_t n_sym, const float* h_weight, float weight0, float* h_llr)
{
    SDR_TRY(ensure_device(ctx));
    if (!valid_mod(mod)) return fail(SDR_ERR_INVALID, "unknown modulation %d", mod);
    if (n_sym == 0) return SDR_OK;
    if (!h_sym || !h_llr) return fail(SDR_ERR_INVALID, "null host pointer");
    // one arena from the context's staging (kept for the next call), like sdr_ofdm_demod_host_c32
    auto up = [](size_t v) { return (v + 255) & ~static_cast<size_t>(255); };
    const size_t o_w = up(n_sym * 8);
    const size_t o_out = o_w + (h_weight ? up(n_sym * 4) : 0);
    void* arena_v = nullptr;
    SDR_TRY(get_staging(ctx, 3, o_out + up(n_sym * mod * 4), &arena_v));
    unsigned char* arena = static_cast<unsigned char*>(arena_v);
    float* d_w = h_weight ? reinterpret_cast<float*>(arena + o_w) : nullptr;
    SDR_HIP(hipMemcpyAsync(arena, h_sym, n_sym * 8, hipMemcpyHostToDevice, ctx->stream));
    if (h_weight) SDR_HIP(hipMemcpyAsync(d_w, h_weight, n_sym * 4, hipMemcpyHostToDevice, ctx->stream));
    SDR_TRY(soft_demap_dev(ctx, mod, arena, n_sym, d_w, weight0, reinterpret_cast<float*>(arena + o_out)));
    SDR_HIP(hipMemcpyAsync(h_llr, arena + o_out, n_sym * mod * 4, hipMemcpyDeviceToHost, ctx->stream));
    SDR_HIP(hipStreamSynchronize(ctx->stream));
    return SDR_OK;
}

int sdr_ofdm_demod_host_c32(sdr_ctx* ctx, const sdr_demod_args* host_args, size_t n_sets)
{
    SDR_TRY(ensure_device(ctx));
    sdr_demod_args a;
    SDR_TRY(read_demod_args(host_args, a));
    const sdr_demod_args given = a;      // the caller's host pointers
    if (a.n_sym == 0 || a.n_fft == 0) return SDR_OK;
    if (!is_pow2(a.n_fft)) return fail(SDR_ERR_NOT_POW2, "%s", kPow2Message);
    if (!a.d_rx) return fail(SDR_ERR_INVALID, "null rx");
    const size_t bins = size_t(a.n_sym) * a.n_fft;
    const size_t rx_bytes = size_t(a.n_sym) * (size_t(a.n_fft) + a.cp) * 8;
    const size_t hard_bytes = bins * a.mod / 8;
    // one arena from the context's staging (grown on demand, kept for the next call) instead of
    // eight hipMalloc / hipFree pairs per call
    // the kernel indexes tap set sym / sym_per_set (set 0 alone when sym_per_set is 0): every set it
    // can reach must be behind the host pointers, or it would read the neighbouring arena regions
    const size_t need_sets = a.sym_per_set ? static_cast<size_t>((a.n_sym + a.sym_per_set - 1) / a.sym_per_set) : 1;
    if ((a.d_eq || a.d_llr_weight) && n_sets < need_sets)
        return fail(SDR_ERR_INVALID, "n_sets = %zu but %llu symbols with sym_per_set = %u index %zu tap / weight sets", n_sets,
                    static_cast<unsigned long long>(a.n_sym), a.sym_per_set, need_sets);
    const size_t set_bytes = n_sets * a.n_fft;
    auto up = [](size_t v) { return (v + 255) & ~static_cast<size_t>(255); };
    const size_t o_rx = 0;
    const size_t o_eq = o_rx + up(rx_bytes);
    const size_t o_wgt = o_eq + (a.d_eq ? up(set_bytes * 8) : 0);
    const size_t o_ref = o_wgt + (a.d_llr_weight ? up(set_bytes * 4) : 0);
    const size_t o_cnt = o_ref + (a.d_ref_hard ? up(hard_bytes) : 0);
    const size_t o_llr = o_cnt + (a.d_counters ? up(64) : 0);
    const size_t o_hard = o_llr + (a.d_llr ? up(bins * a.mod * 4) : 0);
    const size_t o_fft = o_hard + (a.d_hard ? up(hard_bytes) : 0);
    const size_t total = o_fft + (a.d_fft ? up(bins * 8) : 0);
    void* arena_v = nullptr;
    SDR_TRY(get_staging(ctx, 3, total, &arena_v));
    unsigned char* arena = static_cast<unsigned char*>(arena_v);
    struct { void* p; } rx{arena + o_rx}, eq{arena + o_eq}, wgt{arena + o_wgt}, ref{arena + o_ref}, cnt{arena + o_cnt}, llr{arena + o_llr}, hard{arena + o_hard}, fft{arena + o_fft};
    SDR_HIP(hipMemcpyAsync(rx.p, a.d_rx, rx_bytes, hipMemcpyHostToDevice, ctx->stream));
    a.d_rx = rx.p;
    if (a.d_eq) { SDR_HIP(hipMemcpyAsync(eq.p, a.d_eq, set_bytes * 8, hipMemcpyHostToDevice, ctx->stream)); a.d_eq = eq.p; }
    if (a.d_llr_weight) { SDR_HIP(hipMemcpyAsync(wgt.p, a.d_llr_weight, set_bytes * 4, hipMemcpyHostToDevice, ctx->stream)); a.d_llr_weight = static_cast<const float*>(wgt.p); }
    if (a.d_ref_hard) { SDR_HIP(hipMemcpyAsync(ref.p, a.d_ref_hard, hard_bytes, hipMemcpyHostToDevice, ctx->stream)); a.d_ref_hard = static_cast<const uint8_t*>(ref.p); }
    if (a.d_counters) { SDR_HIP(hipMemcpyAsync(cnt.p, a.d_counters, 64, hipMemcpyHostToDevice, ctx->stream)); a.d_counters = static_cast<uint64_t*>(cnt.p); }
    if (a.d_llr) a.d_llr = static_cast<float*>(llr.p);
    if (a.d_hard) a.d_hard = static_cast<uint8_t*>(hard.p);
    if (a.d_fft) a.d_fft = fft.p;
    SDR_TRY(sdr_ofdm_demod_c32(ctx, &a));
    if (given.d_llr) SDR_HIP(hipMemcpyAsync(given.d_llr, llr.p, bins * a.mod * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (given.d_hard) SDR_HIP(hipMemcpyAsync(given.d_hard, hard.p, hard_bytes, hipMemcpyDeviceToHost, ctx->stream));
    if (given.d_fft) SDR_HIP(hipMemcpyAsync(given.d_fft, fft.p, bins * 8, hipMemcpyDeviceToHost, ctx->stream));
    if (given.d_counters) SDR_HIP(hipMemcpyAsync(given.d_counters, cnt.p, 64, hipMemcpyDeviceToHost, ctx->stream));
    SDR_HIP(hipStreamSynchronize(ctx->stream));
    return SDR_OK;
}

int sdr_link_generate_c32(sdr_ctx* ctx, const sdr_link_args* args)
{
    SDR_TRY(ensure_device(ctx));
    if (!args) return fail(SDR_ERR_INVALID, "null args");
    return link_generate(ctx, *args);
}

int sdr_link_ber_c32(sdr_ctx* ctx, const sdr_link_args* args)
{
    SDR_TRY(ensure_device(ctx));
    if (!args) return fail(SDR_ERR_INVALID, "null args");
    return link_ber(ctx, *args);
}

int sdr_link_sweep_c32(sdr_ctx* ctx, const sdr_sweep_args* args)
{
    SDR_TRY(ensure_device(ctx));
    if (!args) return fail(SDR_ERR_INVALID, "null args");
    return link_sweep(ctx, *args);
}

int sdr_link_channel_c32(sdr_ctx* ctx, uint32_t n_fft, uint64_t seed, uint64_t first_set, size_t n_sets, void* d_chan)
{
    SDR_TRY(ensure_device(ctx));
    return link_channel(ctx, n_fft, seed, first_set, n_sets, d_chan);
}

int sdr_link_chain_workspace(const sdr_chain_args* args, uint64_t* bytes)
{
    if (!args || !bytes) return fail(SDR_ERR_INVALID, "null argument");
    return link_chain_workspace(*args, bytes);
}

int sdr_link_chain_c32(sdr_ctx* ctx, const sdr_chain_args* args)
{
    SDR_TRY(ensure_device(ctx));
    if (!args) return fail(SDR_ERR_INVALID, "null args");
    return link_chain(ctx, *args);
}

int sdr_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    if (!ctr || !key || !out) return fail(SDR_ERR_INVALID, "null pointer");
    const Philox4 r = philox4x32_10(ctr[0], ctr[1], ctr[2], ctr[3], key[0], key[1]);
    for (int i = 0; i < 4; ++i) out[i] = r.v[i];
    return SDR_OK;
}

} // extern "C"
