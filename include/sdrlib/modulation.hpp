// modulation.hpp -- drop-in for the reference's lib/inc/modulation.hpp: same tag types, the same
// e16QAM members, the same to_constl / from_constl signatures; decisions and mappings run on
// the GPU through libsdr_amd.  The tags the reference leaves empty (:13-14, :70-72) are filled
// by the same construction rule (see include/sdr_amd.h), and soft outputs are added.
//
//   modulation::e16QAM::{bits_per_symbol, norm<T>, inorm<T>, table<T>, nearest<T>}  reference :15-68
//   modulation::to_constl<Mod, T = double>(bytes, Mod = {})                          reference :78-94
//   modulation::from_constl<Mod, T = double>(symbols, Mod = {})                      reference :96-112
//   modulation::llr<Mod>(symbols, weight)            additive: max-log LLRs, fp32
// Device failures (no GPU, HIP errors) cannot be returned through these value-returning
// signatures; to_constl / from_constl / llr throw std::runtime_error with the library's message.
// e16QAM::nearest is noexcept like the reference's (:50): a device failure there prints the
// library's message to stderr and calls std::terminate -- loud, never a silently wrong index.
#pragma once

#include <array>
#include <cmath>
#include <complex>
#include <concepts>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <exception>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <vector>

#include "device.hpp"

namespace modulation {

namespace detail {

inline sdr_ctx* context_or_throw()
{
    std::string why;
    sdr_ctx* ctx = sdrlib::thread_context::get(why);
    if (!ctx) throw std::runtime_error("sdrlib: " + why);
    return ctx;
}
inline void check(int rc)
{
    if (rc != SDR_OK) throw std::runtime_error(std::string("sdrlib: ") + sdr_last_error());
}
constexpr unsigned gray_to_binary(unsigned g) { unsigned b = 0; for (; g; g >>= 1) b ^= g; return b; }

// Square Gray QAM with BITS bits per symbol: everything the reference's e16QAM exposes.
template <int BITS>
struct square_qam {
    static_assert(BITS >= 2 && BITS % 2 == 0 && BITS <= 10);
    static constexpr std::size_t bits_per_symbol = BITS;
    static constexpr int axis_bits = BITS / 2;
    static constexpr int levels = 1 << axis_bits;
    static constexpr double energy = 2.0 * (double(levels) * levels - 1.0) / 3.0;

    template <typename T> static constexpr std::array<std::complex<T>, (std::size_t(1) << BITS)> make_table()
    {
        std::array<std::complex<T>, (std::size_t(1) << BITS)> t{};
        for (unsigned i = 0; i < (1u << BITS); ++i) {
            const int li = 2 * int(gray_to_binary(i >> axis_bits)) - (levels - 1);
            const int lq = 2 * int(gray_to_binary(i & (levels - 1))) - (levels - 1);
            t[i] = std::complex<T>(T(li), T(lq));
        }
        return t;
    }
    // un-normalised Gray-coded grid, index -> point (reference table :29-47 for BITS = 4)
    template <typename T> static constexpr std::array<std::complex<T>, (std::size_t(1) << BITS)> table = make_table<T>();
};

} // namespace detail

struct eBPSK { static constexpr std::size_t bits_per_symbol = 1; };
struct eQPSK : detail::square_qam<2> {
    template <typename T> static inline const T norm = T{1} / std::sqrt(T{2});
    template <typename T> static inline const T inorm = std::sqrt(T{2});
};
struct e16QAM : detail::square_qam<4> {
    // norm = T{1}/sqrt(T{10}), inorm = sqrt(T{10}), each rounded in T (reference :20, :23)
    template <typename T> static constexpr T norm = T(0x1.43d136248490fp-2);
    template <typename T> static constexpr T inorm = T(0x1.94c583ada5b53p+1);
    // nearest(): scale by inorm, ascending scan with strict '<' (reference :49-67), on the device
    template <typename T> static std::uint8_t nearest(const std::complex<T>& pt) noexcept
    {
        static_assert(std::is_same_v<T, float> || std::is_same_v<T, double>);
        std::uint16_t idx = 0;
        std::string why;
        sdr_ctx* ctx = sdrlib::thread_context::get(why);
        int rc = SDR_ERR_HIP;
        if (ctx) {
            if constexpr (std::is_same_v<T, float>) rc = sdr_nearest_host_c32(ctx, 4, &pt, 1, &idx);
            else rc = sdr_nearest_host_c64(ctx, 4, &pt, 1, &idx);
            if (rc != SDR_OK) why = sdr_last_error();
        }
        if (rc != SDR_OK) {
            std::fprintf(stderr, "sdrlib: e16QAM::nearest cannot reach the device: %s\n", why.c_str());
            std::terminate();
        }
        return static_cast<std::uint8_t>(idx);
    }
};
struct e64QAM : detail::square_qam<6> {
    template <typename T> static inline const T norm = T{1} / std::sqrt(T{42});
    template <typename T> static inline const T inorm = std::sqrt(T{42});
};
struct e256QAM : detail::square_qam<8> {
    template <typename T> static inline const T norm = T{1} / std::sqrt(T{170});
    template <typename T> static inline const T inorm = std::sqrt(T{170});
};
struct e1024QAM : detail::square_qam<10> {
    template <typename T> static inline const T norm = T{1} / std::sqrt(T{682});
    template <typename T> static inline const T inorm = std::sqrt(T{682});
};

template <typename Mod>
concept modulation_tag = std::same_as<Mod, eBPSK> || std::same_as<Mod, eQPSK> || std::same_as<Mod, e16QAM> ||
                         std::same_as<Mod, e64QAM> || std::same_as<Mod, e256QAM> || std::same_as<Mod, e1024QAM>;

// bytes -> constellation points, MSB first in groups of bits_per_symbol (16-QAM: high nibble
// first, reference :87-91); floor(8*bytes/bits) symbols.
template <typename Mod, typename T = double>
std::vector<std::complex<T>> to_constl(const std::vector<std::uint8_t>& in, Mod = {})
    requires modulation_tag<Mod>
{
    static_assert(std::is_same_v<T, float> || std::is_same_v<T, double>);
    const std::size_t n_sym = in.size() * 8 / Mod::bits_per_symbol;
    std::vector<std::complex<T>> out(n_sym);
    if (n_sym == 0) return out;
    sdr_ctx* ctx = detail::context_or_throw();
    if constexpr (std::is_same_v<T, float>) detail::check(sdr_to_constl_host_c32(ctx, int(Mod::bits_per_symbol), in.data(), n_sym, out.data()));
    else detail::check(sdr_to_constl_host_c64(ctx, int(Mod::bits_per_symbol), in.data(), n_sym, out.data()));
    return out;
}

// constellation points -> hard-decision bytes; symbols that do not complete a byte are dropped
// (16-QAM: the trailing unpaired symbol, reference :103).
template <typename Mod, typename T = double>
std::vector<std::uint8_t> from_constl(const std::vector<std::complex<T>>& in, Mod = {})
    requires modulation_tag<Mod>
{
    static_assert(std::is_same_v<T, float> || std::is_same_v<T, double>);
    std::vector<std::uint8_t> out(in.size() * Mod::bits_per_symbol / 8);
    if (out.empty()) return out;
    sdr_ctx* ctx = detail::context_or_throw();
    if constexpr (std::is_same_v<T, float>) detail::check(sdr_from_constl_host_c32(ctx, int(Mod::bits_per_symbol), in.data(), in.size(), out.data()));
    else detail::check(sdr_from_constl_host_c64(ctx, int(Mod::bits_per_symbol), in.data(), in.size(), out.data()));
    return out;
}

// Max-log LLRs (no reference counterpart): bits_per_symbol values per symbol, index MSB first,
// positive = bit 0; see include/sdr_amd.h for the exact definition.  fp32 only.
template <typename Mod>
std::vector<float> llr(const std::vector<std::complex<float>>& in, float weight, Mod = {})
    requires modulation_tag<Mod>
{
    std::vector<float> out(in.size() * Mod::bits_per_symbol);
    if (out.empty()) return out;
    detail::check(sdr_soft_demap_host_c32(detail::context_or_throw(), int(Mod::bits_per_symbol), in.data(), in.size(), nullptr, weight, out.data()));
    return out;
}

} // namespace modulation
