// bitpack.hpp -- compile-time loops and the packing of hard decisions into the MSB-first bit stream of
// modulation::from_constl (lib/inc/modulation.hpp:105-108, extended to any order): shared by the fused receive kernel,
// the frame generator (transmitted bits) and the element-wise from_constl kernel.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>
#include <utility>

namespace sdr {

template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f)
{
    [&]<int... I>(std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }(std::make_integer_sequence<int, N>{});
}

// Packs one symbol's decisions (one idx_t of BPS bits per bin, natural order, in LDS) into the MSB-first bit stream
// (lib/inc/modulation.hpp:105-108 for any order).  One-byte decisions of an even number of bits: a thread takes 16
// consecutive decisions with one 16-byte LDS read, squeezes their 16 * BPS bits into BPS / 2 words in registers and stores
// them with one instruction (a wave writes 64 * 2 * BPS contiguous bytes); the word-by-word form it replaces, still used for
// BPSK and 1024-QAM, read every decision as a single byte (7 LDS reads and ~28 instructions per word at 64-QAM).
// `first` / `step`: the calling thread's index among the threads of the symbol and their number; the caller has made the
// decisions visible (barrier) and `out` is 4-byte aligned.
template <int N, int BPS, typename idx_t>
__device__ __forceinline__ void pack_decisions(const idx_t* __restrict__ dec, uint8_t* __restrict__ out_bytes, int first, int step)
{
    uint32_t* out = reinterpret_cast<uint32_t*>(out_bytes);
    if constexpr (sizeof(idx_t) == 1 && BPS % 2 == 0 && N % 16 == 0) {
        constexpr int NW = BPS / 2;                                   // words per 16 decisions
        for (int c = first; c < N / 16; c += step) {
            const uint4 r4 = *reinterpret_cast<const uint4*>(dec + 16 * c);
            const uint32_t r[4] = {r4.x, r4.y, r4.z, r4.w};
            uint32_t q[4];                                            // 4 decisions = 4 * BPS bits, first decision highest
            static_for<4>([&](auto J) {
                constexpr int j = J;
                q[j] = ((r[j] & 0xffu) << (3 * BPS)) | (((r[j] >> 8) & 0xffu) << (2 * BPS)) | (((r[j] >> 16) & 0xffu) << BPS) | (r[j] >> 24);
            });
            uint32_t wds[NW];
            static_for<NW>([&](auto M_) {
                constexpr int m = M_;
                uint32_t acc = 0;
                static_for<4>([&](auto J) {
                    constexpr int j = J;
                    constexpr int sh = 32 * (m + 1) - 4 * BPS * (j + 1);       // position of q[j]'s lowest bit in word m
                    if constexpr (sh >= 0 && sh < 32) acc |= q[j] << sh;
                    else if constexpr (sh < 0 && sh > -4 * BPS) acc |= q[j] >> (-sh);
                });
                wds[m] = __builtin_bswap32(acc);
            });
            uint32_t* o = out + NW * c;
            if constexpr (NW == 1) o[0] = wds[0];
            else if constexpr (NW == 2) { typedef uint32_t u2 __attribute__((ext_vector_type(2), aligned(4))); u2 v2 = {wds[0], wds[1]}; *reinterpret_cast<u2*>(o) = v2; }
            else if constexpr (NW == 3) { typedef uint32_t u3 __attribute__((ext_vector_type(3), aligned(4))); u3 v3 = {wds[0], wds[1], wds[2]}; *reinterpret_cast<u3*>(o) = v3; }
            else { typedef uint32_t u4 __attribute__((ext_vector_type(4), aligned(4))); u4 v4 = {wds[0], wds[1], wds[2], wds[3]}; *reinterpret_cast<u4*>(o) = v4; }
        }
    } else {
        constexpr int WORDS = N * BPS / 32;
        constexpr int CNT = (32 + BPS - 1) / BPS + 1;
        for (int j = first; j < WORDS; j += step) {
            const int k0 = (32 * j) / BPS;
            const int off = 32 * j - k0 * BPS;
            unsigned long long acc = 0;
#pragma unroll
            for (int i = 0; i < CNT; ++i) {
                const int kk = (k0 + i < N) ? k0 + i : N - 1;
                acc = (acc << BPS) | dec[kk];
            }
            out[j] = __builtin_bswap32(static_cast<uint32_t>(acc >> (CNT * BPS - off - 32)));
        }
    }
}

} // namespace sdr
