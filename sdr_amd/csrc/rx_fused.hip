// rx_fused.hip -- host side of the tuned fused receive kernel: eligibility, argument packing,
// dispatch to the per-size instantiations (rx_fused_inst.hip).
#include "common.hpp"
#include "kernels.hpp"
#include "qam.hpp"
#include "rx_fused.hpp"

#include <cmath>
#include <string>

namespace sdr {

int rx_fused_dispatch_6(sdr_ctx*, const RxParams&, int, int);
int rx_fused_dispatch_7(sdr_ctx*, const RxParams&, int, int);
int rx_fused_dispatch_8(sdr_ctx*, const RxParams&, int, int);
int rx_fused_dispatch_9(sdr_ctx*, const RxParams&, int, int);
int rx_fused_dispatch_10(sdr_ctx*, const RxParams&, int, int);
int rx_fused_dispatch_11(sdr_ctx*, const RxParams&, int, int);
int rx_fused_dispatch_12(sdr_ctx*, const RxParams&, int, int);
int rx_fused_dispatch_13(sdr_ctx*, const RxParams&, int, int);

RxTuning& rx_tuning_mutable()
{
    static RxTuning t;            // C++11: initialised once (from the environment), thread-safe
    return t;
}

int rx_set_tuning(const char* name, long value)
{
    if (!name) return fail(SDR_ERR_INVALID, "null knob name");
    RxTuning& t = rx_tuning_mutable();
    const std::string n(name);
    if (n == "lds_pad") t.lds_pad = value;
    else if (n == "xcd_window") t.xcd_window = value & ~7L;
    else if (n == "xcd_stagger") t.xcd_stagger = value < 0 ? 0 : value;
    else if (n == "chain_form") t.chain_form = value;
    else if (n == "debug") t.debug = value != 0;
    else return fail(SDR_ERR_INVALID, "unknown tuning knob '%s'", name);
    return SDR_OK;
}

namespace {

bool aligned(const void* p, size_t a) { return (reinterpret_cast<uintptr_t>(p) % a) == 0; }

int dispatch(sdr_ctx* ctx, unsigned log2n, const RxParams& p, int mod, int variant)
{
    switch (log2n) {
        case 6: return rx_fused_dispatch_6(ctx, p, mod, variant);
        case 7: return rx_fused_dispatch_7(ctx, p, mod, variant);
        case 8: return rx_fused_dispatch_8(ctx, p, mod, variant);
        case 9: return rx_fused_dispatch_9(ctx, p, mod, variant);
        case 10: return rx_fused_dispatch_10(ctx, p, mod, variant);
        case 11: return rx_fused_dispatch_11(ctx, p, mod, variant);
        case 12: return rx_fused_dispatch_12(ctx, p, mod, variant);
        case 13: return rx_fused_dispatch_13(ctx, p, mod, variant);
        default: return fail(SDR_ERR_UNSUPPORTED, "n_fft = 2^%u has no fused kernel", log2n);
    }
}

} // namespace

// mod = 0 asks about the FFT-only form.
bool rx_fused_supported(uint32_t n_fft, uint32_t mod)
{
    if (!is_pow2(n_fft) || n_fft < 64 || n_fft > 8192) return false;
    return mod == 0 || mod == 1 || mod == 2 || mod == 4 || mod == 6 || mod == 8 || mod == 10;
}

// The vector accesses of the kernel need 16-byte aligned buffers and an even CP; anything else
// goes to the generic engines (same arithmetic).
bool rx_fused_layout_ok(const sdr_demod_args& a)
{
    if (a.cp % 2 != 0) return false;
    if (!aligned(a.d_rx, 16) || !aligned(a.d_llr, 16) || !aligned(a.d_fft, 16) || !aligned(a.d_eq, 16) || !aligned(a.d_llr_weight, 8)) return false;
    if (!aligned(a.d_hard, 4) || !aligned(a.d_ref_hard, 4)) return false;
    return true;
}

int rx_fused_launch(sdr_ctx* ctx, const sdr_demod_args& a, const RxWorkspace* ws)
{
    const void* tw = ws ? ws->twiddles : nullptr;
    if (!tw) SDR_TRY(get_twiddles(ctx, a.n_fft, false, false, &tw));
    RxParams p = {};
    p.rx = static_cast<const float2*>(a.d_rx);
    p.tw = static_cast<const float2*>(tw);
    p.eq = static_cast<const float2*>(a.d_eq);
    p.wgt = a.d_llr_weight;
    p.llr = a.d_llr;
    p.hard = a.d_hard;
    p.fft = static_cast<float2*>(a.d_fft);
    p.ref = a.d_ref_hard;
    if (a.d_counters) {
        if (ws && ws->counter_slots) p.counters = ws->counter_slots;
        else SDR_TRY(get_counter_slots(ctx, &p.counters));
    }
    p.n_sym = a.n_sym;
    p.cp = a.cp;
    p.sym_per_set = a.sym_per_set;
    p.w0 = a.llr_weight0;
    p.w_from_taps = (a.flags & SDR_DEMOD_WEIGHT_FROM_TAPS) && a.d_eq && !a.d_llr_weight;
    p.inorm = std::sqrt(static_cast<float>(grid_energy(static_cast<int>(a.mod))));
    const bool full = a.d_hard || a.d_fft || a.d_counters || !a.d_llr;
    const bool hard_only = a.d_hard && !a.d_llr && !a.d_fft && !a.d_counters && !a.d_eq && a.mod != 1;   // weights do not matter to nearest()
    int variant = hard_only ? 4 : (full ? 2 : 0) | ((a.d_eq || a.d_llr_weight) ? 1 : 0);
    if (variant == 1 && !a.d_llr_weight) variant = 9;      // LLRs only, taps without a weight array: no weight loads at all
    if (variant == 1 && !a.d_eq) variant = 3;              // weights without taps: the every-output kernel tests its arrays at run time (the LLR-only one assumes both)
    // LLRs + counters and nothing else: the every-output kernel with its outputs fixed at compile time (RxShape LC).  Same-buffer
    // A/B against the run-time form (profiles/r3_llr_counters_instantiation_ab.txt): +1.4 to +5.0 points without taps up to
    // n_fft 2048 and 64-QAM; with taps, from 256-QAM or from n_fft 4096 on the straight-line code needs 17 more registers,
    // loses a wave per SIMD and 4.6 to 17.5 points -- those keep the run-time form.
    if (variant == 2 && a.d_llr && a.d_counters && a.d_ref_hard && !a.d_hard && !a.d_fft && a.mod >= 2 && a.mod <= 6 && a.n_fft <= 2048) variant = 10;
    SDR_TRY(dispatch(ctx, ilog2(a.n_fft), p, static_cast<int>(a.mod), variant));
    if (a.d_counters) SDR_TRY(fold_counter_slots(ctx, p.counters, reinterpret_cast<unsigned long long*>(a.d_counters), a.n_sym * a.n_fft, a.mod, a.n_sym));
    return SDR_OK;
}

int rx_fft_only_launch(sdr_ctx* ctx, const void* d_in, uint32_t n_fft, uint32_t cp, uint64_t n_sym, void* d_out)
{
    const void* tw = nullptr;
    SDR_TRY(get_twiddles(ctx, n_fft, false, false, &tw));
    RxParams p = {};
    p.rx = static_cast<const float2*>(d_in);
    p.tw = static_cast<const float2*>(tw);
    p.fft = static_cast<float2*>(d_out);
    p.n_sym = n_sym;
    p.cp = cp;
    return dispatch(ctx, ilog2(n_fft), p, 0, 2);
}

// ---- diagnostic: the memory system's ceiling for this kernel's traffic shape -------------------
// Same access pattern as the fused receive kernel and no arithmetic: one wave per symbol reads
// the symbol's n_fft samples (cyclic prefix skipped) with 16-byte loads, then writes the symbol's
// n_fft * bps floats with lane-contiguous 16-byte non-temporal stores; one symbol per
// workgroup; the same XCD-aware workgroup order.  bench.py reports its rate beside the
// kernel's, so that "fraction of the HBM peak" can be read against what a pure copy of this
// shape reaches on the same box.
namespace {

__global__ void __launch_bounds__(64) copy_shape_kernel(const float4* __restrict__ rx, float* __restrict__ out, unsigned n_fft, unsigned cp,
                                                        unsigned bps, unsigned long long n_sym, unsigned xcd_window)
{
    unsigned long long wg = blockIdx.x;
    if (xcd_window >= 8) {
        const unsigned long long win = xcd_window;
        const unsigned long long full = (static_cast<unsigned long long>(gridDim.x) / win) * win;
        if (wg < full) {
            const unsigned long long r = wg % win;
            wg = wg - r + (r & 7) * (win >> 3) + (r >> 3);
        }
    }
    const unsigned lane = threadIdx.x;
    const unsigned in_vec = n_fft / 2;                    // float4 = two samples
    const unsigned out_vec = n_fft * bps / 4;             // float4 = four LLRs
    for (unsigned long long sym = wg; sym < wg + 1 && sym < n_sym; ++sym) {           // one symbol per workgroup, like the kernel
        const float4* src = rx + (sym * (static_cast<unsigned long long>(n_fft) + cp) + cp) / 2;
        float acc = 0.0f;
        for (unsigned i = lane; i < in_vec; i += 64) {
            const float4 v = src[i];
            acc += v.x + v.y + v.z + v.w;
        }
        float* dst = out + sym * static_cast<unsigned long long>(n_fft) * bps;
        for (unsigned i = lane; i < out_vec; i += 64) stream_store4(dst + 4ull * i, acc, acc, acc, acc);
    }
}

} // namespace

int copy_shape_launch(sdr_ctx* ctx, const void* d_rx, uint32_t n_fft, uint32_t cp, uint32_t bps, uint64_t n_sym, float* d_out)
{
    if (n_fft < 64 || (n_fft & (n_fft - 1)) != 0 || cp % 2 != 0 || bps == 0 || (static_cast<uint64_t>(n_fft) * bps) % 4 != 0)
        return fail(SDR_ERR_INVALID, "copy-shape diagnostic: n_fft >= 64 power of two, even cp");
    if (!aligned(d_rx, 16) || !aligned(d_out, 16) || !d_rx || !d_out) return fail(SDR_ERR_INVALID, "copy-shape diagnostic: 16-byte aligned buffers");
    const unsigned long long blocks = n_sym;
    if (blocks == 0) return SDR_OK;
    if (blocks > 0x7FFFFFFFull) return fail(SDR_ERR_UNSUPPORTED, "batch too large for one launch");
    const RxTuning& tune = rx_tuning();
    hipLaunchKernelGGL(copy_shape_kernel, dim3(static_cast<unsigned>(blocks)), dim3(64), 0, ctx->stream, static_cast<const float4*>(d_rx), d_out,
                       n_fft, cp, bps, static_cast<unsigned long long>(n_sym),
                       static_cast<unsigned>(tune.xcd_window > 0 ? (static_cast<unsigned long long>(tune.xcd_window) < (blocks & ~7ull) ? static_cast<unsigned long long>(tune.xcd_window) : (blocks & ~7ull)) : 0));      // same clipped window as the receive kernel
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail_hip(e, "copy_shape_kernel launch");
    return SDR_OK;
}

} // namespace sdr
