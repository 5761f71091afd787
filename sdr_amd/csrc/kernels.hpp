// kernels.hpp -- internal entry points of the HIP translation units (not part of the C ABI).
#pragma once

#include "common.hpp"

namespace sdr {

// fft_generic.hip: the reference network for any power-of-two length, fp32/fp64.
int fft_generic_f32(sdr_ctx* ctx, const void* d_in, void* d_out, size_t n, bool inverse,
                    size_t in_stride, size_t in_offset, size_t out_stride, size_t out_offset, size_t cp_fill, size_t n_seq);
int fft_generic_f64(sdr_ctx* ctx, const void* d_in, void* d_out, size_t n, bool inverse,
                    size_t in_stride, size_t in_offset, size_t out_stride, size_t out_offset, size_t cp_fill, size_t n_seq);

// demap.hip: element-wise constellation kernels.
int to_constl_dev(sdr_ctx* ctx, bool is_double, int mod, const uint8_t* d_bytes, size_t n_sym, void* d_out);
int from_constl_dev(sdr_ctx* ctx, bool is_double, int mod, const void* d_sym, size_t n_sym, uint8_t* d_bytes);
int nearest_dev(sdr_ctx* ctx, bool is_double, int mod, const void* d_sym, size_t n_sym, uint16_t* d_index);
int soft_demap_dev(sdr_ctx* ctx, int mod, const void* d_sym, size_t n_sym, const float* d_weight, float weight0, float* d_llr);
int zf_taps_dev(sdr_ctx* ctx, int mod, const void* d_chan, size_t n, float inv_n0, void* d_taps, float* d_weight);
// equalise + LLR + hard + counters on FFT outputs already in memory (generic sizes of the fused path)
int demap_after_fft(sdr_ctx* ctx, const sdr_demod_args& a, const void* d_fft);

// capi.hip: copies the caller's sdr_demod_args (current layout, or the 96-byte round-2 layout without `flags`) and checks
// what every receive entry point checks first
int read_demod_args(const sdr_demod_args* in, sdr_demod_args& out);

// rx_fused.hip: the tuned fused receive kernel (fp32, n_fft 64..8192).
bool rx_fused_supported(uint32_t n_fft, uint32_t mod);
bool rx_fused_layout_ok(const sdr_demod_args& a);
// Device memory the launch needs beyond inputs and outputs.  Null members are taken from the
// context's own (allocating) caches; the allocation-free entry point brings its static array.
struct RxWorkspace {
    const void* twiddles = nullptr;          // n_fft - 1 forward twiddles
    unsigned long long* counter_slots = nullptr;   // kCounterSlots x 8, all zero
};
int rx_fused_launch(sdr_ctx* ctx, const sdr_demod_args& a, const RxWorkspace* ws = nullptr);
// noalloc.hip: resolves the static scratch array of the context's device (once per device; a failure is kept for
// sdr_ofdm_demod_noalloc_c32 to report and does not fail sdr_create) and claims a counter-slot set for the context
void noalloc_attach(sdr_ctx* ctx);
void noalloc_detach(sdr_ctx* ctx);
int rx_fft_only_launch(sdr_ctx* ctx, const void* d_in, uint32_t n_fft, uint32_t cp, uint64_t n_sym, void* d_out);
// measurement aid (sdr_amd_diag.h): changes a run-time tuning knob of the receive kernels
int rx_set_tuning(const char* name, long value);
int copy_shape_launch(sdr_ctx* ctx, const void* d_rx, uint32_t n_fft, uint32_t cp, uint32_t bps, uint64_t n_sym, float* d_out);

// link.hip: frame generator, AWGN, Monte-Carlo link.
// skip_prefix: the cyclic prefix of every symbol is left unwritten (for a consumer that drops it: the chain's receive kernel)
int link_generate(sdr_ctx* ctx, const sdr_link_args& a, bool skip_prefix = false);
int link_ber(sdr_ctx* ctx, const sdr_link_args& a);
int link_sweep(sdr_ctx* ctx, const sdr_sweep_args& a);
int link_channel(sdr_ctx* ctx, uint32_t n_fft, uint64_t seed, uint64_t first_set, size_t n_sets, void* d_chan);
int link_chain_workspace(const sdr_chain_args& a, uint64_t* bytes);
int link_chain(sdr_ctx* ctx, const sdr_chain_args& a);
bool tx_fused_ok(size_t n_fft, size_t cp, const void* d_in, const void* d_out);
int tx_fused_launch(sdr_ctx* ctx, const void* d_in, uint32_t n_fft, uint32_t cp, uint64_t n_sym, void* d_out);

} // namespace sdr
