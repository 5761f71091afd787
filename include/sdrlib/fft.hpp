// fft.hpp -- drop-in for the reference's lib/inc/fft.hpp: same namespace, names, template
// parameters and error behaviour; the transform itself runs on the GPU through libsdr_amd
// (sdr_fft2_host_c32 / _c64), which evaluates the reference's radix-2 DIT network and twiddle
// recurrence (lib/inc/fft.hpp:87-155) operation for operation.
//
//   fft::fft2<PT = 1024>(begin, end)   -> expected<void, std::string>     (reference :176-180)
//   fft::ifft2<PT = 1024>(begin, end)  -> expected<void, std::string>     (reference :194-205)
//   fft::fft_compatible_iterator<It>                                      (reference :161-162)
//
// Kept from the reference: in place; sizes 0 and 1 succeed untouched (:91); any other size
// that is not a power of two fails with "The sequence size must be of powers of 2" (:92);
// ifft2 divides by the length (:200-202).  ParallelThreshold is accepted and ignored (in the
// reference it only feeds an OpenMP pragma that is never active, :119).
// std::complex<float> and std::complex<double> are supported (the device has both paths).
#pragma once

#include <complex>
#include <concepts>
#include <cstddef>
#include <iterator>
#include <string>
#include <type_traits>
#include <vector>

#include "device.hpp"
#include "expected.hpp"

namespace fft::detail {

template <typename T> struct is_complex_t : std::false_type {};
template <typename T> struct is_complex_t<std::complex<T>> : std::true_type {};
template <typename T> concept is_complex = is_complex_t<T>::value;

template <typename It>
concept fft_compatible_iterator = std::random_access_iterator<It> && is_complex<std::iter_value_t<It>>;

template <typename Real>
inline int run_host(sdr_ctx* ctx, std::complex<Real>* data, std::size_t n, bool inverse)
{
    static_assert(std::is_same_v<Real, float> || std::is_same_v<Real, double>, "the device path computes in fp32 or fp64");
    if constexpr (std::is_same_v<Real, float>) return sdr_fft2_host_c32(ctx, data, n, 1, inverse ? 1 : 0);
    else return sdr_fft2_host_c64(ctx, data, n, 1, inverse ? 1 : 0);
}

template <fft_compatible_iterator It>
sdrlib::expected<void, std::string> transform(It begin, It end, bool inverse)
{
    using value_type = std::iter_value_t<It>;
    using real_type = typename value_type::value_type;
    const std::size_t size = static_cast<std::size_t>(std::distance(begin, end));
    if (size > 0 && (size & (size - 1)) != 0) return sdrlib::unexpected<std::string>("The sequence size must be of powers of 2");
    if (size <= 1) return {};
    std::string why;
    sdr_ctx* ctx = sdrlib::thread_context::get(why);
    if (!ctx) return sdrlib::unexpected<std::string>(why);
    int rc;
    if constexpr (std::contiguous_iterator<It>) {
        rc = run_host<real_type>(ctx, std::to_address(begin), size, inverse);
    } else {
        std::vector<value_type> tmp(begin, end);          // e.g. std::deque: gather, transform, scatter
        rc = run_host<real_type>(ctx, tmp.data(), size, inverse);
        if (rc == SDR_OK) std::copy(tmp.begin(), tmp.end(), begin);
    }
    if (rc != SDR_OK) return sdrlib::unexpected<std::string>(sdr_last_error());
    return {};
}

} // namespace fft::detail

namespace fft {

template <typename It>
concept fft_compatible_iterator = detail::fft_compatible_iterator<It>;

template <std::size_t ParallelThreshold = 1024, fft_compatible_iterator It>
sdrlib::expected<void, std::string> fft2(It begin, It end)
{
    return detail::transform(begin, end, false);
}

template <std::size_t ParallelThreshold = 1024, fft_compatible_iterator It>
sdrlib::expected<void, std::string> ifft2(It begin, It end)
{
    return detail::transform(begin, end, true);
}

} // namespace fft
