// ofdm.hpp -- drop-in for the reference's lib/inc/ofdm.hpp (same names and signatures), plus
// batched forms.  All transforms run on the GPU through libsdr_amd.
//
//   ofdm::tx(in, cp, out) / ofdm::tx(in, cp)       reference :13-35
//   ofdm::rx(in, cp, out) / ofdm::rx(in, cp)       reference :37-54
//   ofdm::rx_batch / ofdm::tx_batch                additive: the reference handles ONE OFDM
//                                                  symbol per call (:40-42)
//   ofdm::demod_batch<Mod> / ofdm::zf_equaliser<Mod>   additive: the fused receive path (CP removal -> FFT
//                                                  -> one-tap equalise -> max-log LLR + nearest() hard bits),
//                                                  which sits between ofdm::rx (:45-54) and
//                                                  modulation::from_constl (modulation.hpp:96-112)
// Kept from the reference: the callee resizes `out`; on a length that is not a power of two the
// error text is the FFT's and `out` already holds the copied samples (:16-18, :40-42).
// Deliberate difference: rx with in.size() < cp returns an error instead of the reference's
// size_t underflow (:40).
#pragma once

#include <algorithm>
#include <complex>
#include <concepts>
#include <cstddef>
#include <cstdint>
#include <string>
#include <type_traits>
#include <vector>

#include "device.hpp"
#include "expected.hpp"
#include "fft.hpp"
#include "modulation.hpp"

namespace ofdm {

namespace detail {
inline bool pow2_or_empty(std::size_t n) { return n == 0 || (n & (n - 1)) == 0; }
inline sdrlib::unexpected<std::string> pow2_error() { return sdrlib::unexpected<std::string>("The sequence size must be of powers of 2"); }
}

// Batched receive: `in` holds n_sym blocks of (n_fft + cp) samples; `out` gets n_sym * n_fft bins.
template <std::floating_point T>
sdrlib::expected<void, std::string> rx_batch(const std::vector<std::complex<T>>& in, std::size_t n_fft, std::size_t cp_size, std::vector<std::complex<T>>& out)
{
    static_assert(std::is_same_v<T, float> || std::is_same_v<T, double>, "the device path computes in fp32 or fp64");
    const std::size_t block = n_fft + cp_size;
    if (block == 0 || in.size() % block != 0) return sdrlib::unexpected<std::string>("input length is not a multiple of n_fft + cp");
    const std::size_t n_sym = in.size() / block;
    out.resize(n_sym * n_fft);
    if (!detail::pow2_or_empty(n_fft)) return detail::pow2_error();
    if (n_sym == 0 || n_fft == 0) return {};
    std::string why;
    sdr_ctx* ctx = sdrlib::thread_context::get(why);
    if (!ctx) return sdrlib::unexpected<std::string>(why);
    int rc;
    if constexpr (std::is_same_v<T, float>) rc = sdr_ofdm_rx_host_c32(ctx, in.data(), n_fft, cp_size, n_sym, out.data());
    else rc = sdr_ofdm_rx_host_c64(ctx, in.data(), n_fft, cp_size, n_sym, out.data());
    if (rc != SDR_OK) return sdrlib::unexpected<std::string>(sdr_last_error());
    return {};
}

// Batched transmit: `in` holds n_sym blocks of n_fft bins; `out` gets n_sym * (n_fft + cp) samples.
template <std::floating_point T>
sdrlib::expected<void, std::string> tx_batch(const std::vector<std::complex<T>>& in, std::size_t n_fft, std::size_t cp_size, std::vector<std::complex<T>>& out)
{
    static_assert(std::is_same_v<T, float> || std::is_same_v<T, double>, "the device path computes in fp32 or fp64");
    if (n_fft == 0 || in.size() % n_fft != 0) return sdrlib::unexpected<std::string>("input length is not a multiple of n_fft");
    if (cp_size > n_fft) return sdrlib::unexpected<std::string>("batched tx needs cp <= n_fft");
    const std::size_t n_sym = in.size() / n_fft;
    out.resize(n_sym * (n_fft + cp_size));
    if (!detail::pow2_or_empty(n_fft)) return detail::pow2_error();
    if (n_sym == 0) return {};
    std::string why;
    sdr_ctx* ctx = sdrlib::thread_context::get(why);
    if (!ctx) return sdrlib::unexpected<std::string>(why);
    int rc;
    if constexpr (std::is_same_v<T, float>) rc = sdr_ofdm_tx_host_c32(ctx, in.data(), n_fft, cp_size, n_sym, out.data());
    else rc = sdr_ofdm_tx_host_c64(ctx, in.data(), n_fft, cp_size, n_sym, out.data());
    if (rc != SDR_OK) return sdrlib::unexpected<std::string>(sdr_last_error());
    return {};
}

template <std::floating_point T>
sdrlib::expected<void, std::string> tx(const std::vector<std::complex<T>>& in, std::size_t cp_size, std::vector<std::complex<T>>& out)
{
    const std::size_t n = in.size();
    out.resize(n + cp_size);
    if (!detail::pow2_or_empty(n) || cp_size > n || n <= 1) {
        // shapes outside the batched kernel: place the payload, inverse-transform it in place,
        // then prefix the tail -- the reference's own sequence of steps (:16-21)
        std::copy(in.begin(), in.end(), out.begin() + static_cast<std::ptrdiff_t>(cp_size));
        auto r = fft::ifft2(out.begin() + static_cast<std::ptrdiff_t>(cp_size), out.end());
        if (!r) return r;
        std::copy(out.end() - static_cast<std::ptrdiff_t>(cp_size), out.end(), out.begin());
        return {};
    }
    return tx_batch(in, n, cp_size, out);
}

template <std::floating_point T>
sdrlib::expected<std::vector<std::complex<T>>, std::string> tx(const std::vector<std::complex<T>>& in, std::size_t cp_size)
{
    std::vector<std::complex<T>> out;
    auto r = tx(in, cp_size, out);
    if (!r) return sdrlib::unexpected<std::string>(r.error());
    return out;
}

template <std::floating_point T>
sdrlib::expected<void, std::string> rx(const std::vector<std::complex<T>>& in, std::size_t cp_size, std::vector<std::complex<T>>& out)
{
    if (in.size() < cp_size) return sdrlib::unexpected<std::string>("ofdm::rx: input shorter than the cyclic prefix");
    const std::size_t n = in.size() - cp_size;
    if (!detail::pow2_or_empty(n) || n <= 1) {
        out.resize(n);
        std::copy(in.begin() + static_cast<std::ptrdiff_t>(cp_size), in.end(), out.begin());
        return fft::fft2(out.begin(), out.end());
    }
    return rx_batch(in, n, cp_size, out);
}

template <std::floating_point T>
sdrlib::expected<std::vector<std::complex<T>>, std::string> rx(const std::vector<std::complex<T>>& in, std::size_t cp_size)
{
    std::vector<std::complex<T>> out;
    auto r = rx(in, cp_size, out);
    if (!r) return sdrlib::unexpected<std::string>(r.error());
    return out;
}

// Additive: the fused receive path on a batch of symbols in host memory (CP removal -> FFT ->
// [one-tap equalise ->] max-log LLR + nearest() hard bits in ONE kernel).
struct demod_result {
    std::vector<float> llr;            // n_sym * n_fft * bits, subcarrier-major, index MSB first
    std::vector<std::uint8_t> hard;    // n_sym * n_fft * bits / 8 bytes, MSB first
};

// Per-subcarrier equaliser: one set of n_fft taps G_k and LLR weights w_k per `sym_per_set` consecutive OFDM
// symbols (0 = a single set for the whole batch).  Either vector may be empty (no taps: flat channel; no
// weights: the scalar weight of the call).
struct equaliser {
    std::vector<std::complex<float>> taps;     // n_sets * n_fft, yhat_k = y_k * G_k
    std::vector<float> weights;                // n_sets * n_fft, LLR_b *= w_k
    std::size_t sym_per_set = 0;
};

// Zero-forcing taps and LLR weights from channel estimates H (n_sets * n_fft values):
//   p = |H|^2, G = conj(H) / p, w = norm^2 * p * inv_n0   (include/sdr_amd.h, sdr_zf_taps_c32)
template <typename Mod>
sdrlib::expected<equaliser, std::string> zf_equaliser(const std::vector<std::complex<float>>& chan, float inv_n0, std::size_t sym_per_set, Mod = {})
    requires modulation::modulation_tag<Mod>
{
    equaliser e;
    e.sym_per_set = sym_per_set;
    e.taps.resize(chan.size());
    e.weights.resize(chan.size());
    if (chan.empty()) return e;
    std::string why;
    sdr_ctx* ctx = sdrlib::thread_context::get(why);
    if (!ctx) return sdrlib::unexpected<std::string>(why);
    if (sdr_zf_taps_host_c32(ctx, int(Mod::bits_per_symbol), chan.data(), chan.size(), inv_n0, e.taps.data(), e.weights.data()) != SDR_OK)
        return sdrlib::unexpected<std::string>(sdr_last_error());
    return e;
}

namespace detail {
template <typename Mod>
sdrlib::expected<demod_result, std::string> demod_host(const std::vector<std::complex<float>>& in, std::size_t n_fft, std::size_t cp_size, float llr_weight,
                                                       const equaliser* eq)
{
    const std::size_t block = n_fft + cp_size;
    if (block == 0 || in.size() % block != 0) return sdrlib::unexpected<std::string>("input length is not a multiple of n_fft + cp");
    if (!pow2_or_empty(n_fft)) return pow2_error();
    const std::size_t n_sym = in.size() / block;
    std::size_t n_sets = 0;
    if (eq && (!eq->taps.empty() || !eq->weights.empty())) {
        if (n_fft == 0) return sdrlib::unexpected<std::string>("equaliser given for n_fft = 0");
        const std::size_t len = eq->taps.empty() ? eq->weights.size() : eq->taps.size();
        if (len % n_fft != 0 || (!eq->taps.empty() && !eq->weights.empty() && eq->taps.size() != eq->weights.size()))
            return sdrlib::unexpected<std::string>("equaliser taps / weights must hold whole sets of n_fft values, the same number of each");
        n_sets = len / n_fft;
    }
    demod_result r;
    r.llr.resize(n_sym * n_fft * Mod::bits_per_symbol);
    r.hard.resize(n_sym * n_fft * Mod::bits_per_symbol / 8);
    if (n_sym == 0 || n_fft == 0) return r;
    std::string why;
    sdr_ctx* ctx = sdrlib::thread_context::get(why);
    if (!ctx) return sdrlib::unexpected<std::string>(why);
    sdr_demod_args a{};
    a.struct_size = sizeof(a);
    a.n_fft = static_cast<std::uint32_t>(n_fft);
    a.cp = static_cast<std::uint32_t>(cp_size);
    a.mod = static_cast<std::uint32_t>(Mod::bits_per_symbol);
    a.n_sym = n_sym;
    a.llr_weight0 = llr_weight;
    a.d_rx = in.data();
    a.d_llr = r.llr.data();
    a.d_hard = r.hard.empty() ? nullptr : r.hard.data();
    if (n_sets) {
        a.sym_per_set = static_cast<std::uint32_t>(eq->sym_per_set);
        a.d_eq = eq->taps.empty() ? nullptr : eq->taps.data();
        a.d_llr_weight = eq->weights.empty() ? nullptr : eq->weights.data();
    }
    // too few sets for n_sym / sym_per_set is refused by the library (SDR_ERR_INVALID), not read out of bounds
    if (sdr_ofdm_demod_host_c32(ctx, &a, n_sets) != SDR_OK) return sdrlib::unexpected<std::string>(sdr_last_error());
    return r;
}
} // namespace detail

// flat channel, scalar LLR weight
template <typename Mod>
sdrlib::expected<demod_result, std::string> demod_batch(const std::vector<std::complex<float>>& in, std::size_t n_fft, std::size_t cp_size, float llr_weight, Mod = {})
    requires modulation::modulation_tag<Mod>
{
    return detail::demod_host<Mod>(in, n_fft, cp_size, llr_weight, nullptr);
}

// FFT -> equalise -> LLR: per-subcarrier taps and / or weights, one set per eq.sym_per_set symbols;
// llr_weight is used only when eq.weights is empty
template <typename Mod>
sdrlib::expected<demod_result, std::string> demod_batch(const std::vector<std::complex<float>>& in, std::size_t n_fft, std::size_t cp_size, const equaliser& eq,
                                                        float llr_weight = 1.0f, Mod = {})
    requires modulation::modulation_tag<Mod>
{
    return detail::demod_host<Mod>(in, n_fft, cp_size, llr_weight, &eq);
}

} // namespace ofdm
