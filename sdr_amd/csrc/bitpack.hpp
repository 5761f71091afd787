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
// them with one instruction (a wave writes 64 * 2 * BPS contiguous bytes); BPSK and the two-byte decisions of 1024-QAM have forms of
// their own below; the word-by-word form they replace (any other shape) read every decision by itself (7 LDS reads and ~28
// instructions per word at 64-QAM).
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
    } else if constexpr (sizeof(idx_t) == 1 && BPS == 1 && N % 32 == 0) {
        // BPSK: 32 one-byte decisions (0 / 1) from two 16-byte reads make one word.  Four decisions b0 ... b3 in a register
        // (b0 lowest byte) become the nibble b0 b1 b2 b3 with one multiplication: r * (2^27 + 2^18 + 2^9 + 1) puts b0 ... b3 at bits
        // 27 ... 24 and every other product elsewhere (no two terms share a bit, so nothing carries).
        for (int c = first; c < N / 32; c += step) {
            const uint4 lo = *reinterpret_cast<const uint4*>(dec + 32 * c);
            const uint4 hi = *reinterpret_cast<const uint4*>(dec + 32 * c + 16);
            const uint32_t r[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
            uint32_t acc = 0;
            static_for<8>([&](auto J) {
                constexpr int j = J;
                acc |= (((r[j] & 0x01010101u) * 0x08040201u >> 24) & 0xfu) << (28 - 4 * j);
            });
            out[c] = __builtin_bswap32(acc);
        }
    } else if constexpr (sizeof(idx_t) == 2 && BPS > 8 && BPS <= 16 && (16 * BPS) % 32 == 0 && N % 16 == 0) {
        // two-byte decisions (1024-QAM): 16 of them from two 16-byte reads, 16 * BPS bits = BPS / 2 words through a 64-bit window
        constexpr int NW = BPS / 2;
        for (int c = first; c < N / 16; c += step) {
            const uint4 lo = *reinterpret_cast<const uint4*>(dec + 16 * c);
            const uint4 hi = *reinterpret_cast<const uint4*>(dec + 16 * c + 8);
            const uint32_t r[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
            uint32_t wds[NW];
            unsigned long long acc = 0;
            static_for<16>([&](auto K) {
                constexpr int k = K;
                const uint32_t d = (k & 1) ? (r[k >> 1] >> 16) : (r[k >> 1] & 0xffffu);
                acc = (acc << BPS) | d;
                constexpr int have = (k + 1) * BPS;                   // bits gathered so far
                constexpr int done = (k * BPS) / 32;                  // words emitted before this decision
                if constexpr (have / 32 > done) {
                    constexpr int rest = have - 32 * (done + 1);      // bits of the window that belong to the next word
                    wds[done] = __builtin_bswap32(static_cast<uint32_t>(acc >> rest));
                }
            });
            uint32_t* o = out + NW * c;
            static_for<NW>([&](auto M_) { constexpr int m = M_; o[m] = wds[m]; });
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
