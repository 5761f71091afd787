// rng.hpp -- counter-based randomness of the device frame generator / AWGN channel.
// No reference counterpart.  Philox4x32-10 (Salmon et al., SC'11) plus
// elementary functions written with +,-,* only, one correctly rounded sqrt and two small tables of
// correctly rounded doubles, so that a CPU implementation of the same formulas yields identical bits
// (the build uses -ffp-contract=off).  Round 3: 32-bit uniform for the radius (tail to 6.66 sigma instead
// of 5.77), table-based range reduction in ln and sincos (a third of the operations), two prefix samples
// per Philox block.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sdr {

constexpr uint32_t kStreamTx = 0x54580000u;      // transmitted constellation indices
constexpr uint32_t kStreamNoise = 0x4E5A0000u;   // noise on payload samples
constexpr uint32_t kStreamCp = 0x43500000u;      // noise on cyclic-prefix samples
constexpr uint32_t kStreamChan = 0x43480000u;    // channel realisations (one per set of symbols)

struct Philox4 { uint32_t v[4]; };

__host__ __device__ inline Philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1)
{
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = static_cast<uint64_t>(0xD2511F53u) * c0;
        const uint64_t p1 = static_cast<uint64_t>(0xCD9E8D57u) * c2;
        const uint32_t n0 = static_cast<uint32_t>(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n1 = static_cast<uint32_t>(p1);
        const uint32_t n2 = static_cast<uint32_t>(p0 >> 32) ^ c3 ^ k1;
        const uint32_t n3 = static_cast<uint32_t>(p0);
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    Philox4 out;
    out.v[0] = c0; out.v[1] = c1; out.v[2] = c2; out.v[3] = c3;
    return out;
}

// Block of 8 sixteen-bit lanes serving subcarriers {r, r+stride, ..}, stride = max(N/8, 1).
__device__ inline Philox4 tx_block(uint64_t seed, uint64_t g, uint32_t r)
{
    return philox4x32_10(r, static_cast<uint32_t>(g), static_cast<uint32_t>(g >> 32), kStreamTx,
                         static_cast<uint32_t>(seed), static_cast<uint32_t>(seed >> 32));
}
__device__ inline uint32_t tx_lane(const Philox4& b, uint32_t lane, int bps)
{
    const uint32_t word = (lane & 4u) ? ((lane & 2u) ? b.v[3] : b.v[2]) : ((lane & 2u) ? b.v[1] : b.v[0]);
    return ((word >> (16u * (lane & 1u))) & 0xFFFFu) & ((1u << bps) - 1u);
}
__device__ inline uint32_t tx_index(uint64_t seed, uint64_t g, uint32_t k, uint32_t n_fft, int bps)
{
    const uint32_t stride = n_fft >= 8 ? n_fft / 8 : 1;
    return tx_lane(tx_block(seed, g, k % stride), (k / stride) & 7u, bps);
}

// Tables of the Gaussian generator (round 3), built on the host by make_rng_tables (context.hip) with the same
// double-precision formulas a CPU restatement uses: kRngLn pairs (r_i, L_i) -- r_i = fl(1/c_i), L_i = fl(-ln r_i) for the centre
// c_i = 1 + (i + 1/2)/32 of mantissa interval i -- followed by kRngSc pairs (sin, cos)(2 pi j / 256).  Kernels keep them
// in LDS (fused kernels) or read them from device memory (generic kernels): `Tab` is any pointer to the float2 array.
constexpr int kRngLn = 32, kRngSc = 256, kRngPairs = kRngLn + kRngSc;
constexpr int kRngBytes = kRngPairs * 8;      // 2304

// ln(x) for x in [2^-32, 1]: ln m = L_i + log1p(m r_i - 1), |m r_i - 1| < 1/63, log1p cut after the cubic term (< 2e-8)
template <typename Tab>
__device__ inline float ln_unit(float x, Tab tab)
{
    const uint32_t bits = __float_as_uint(x);
    const int e = static_cast<int>(bits >> 23) - 127;
    const float m = __uint_as_float((bits & 0x007FFFFFu) | 0x3F800000u);
    const float2 rl = tab[(bits >> 18) & 31u];
    const float f = m * rl.x - 1.0f;
    const float q = f * 0.333333343f + -0.5f;
    const float z = f * f;
    const float p = z * q + f;
    const float fe = static_cast<float>(e);
    return (fe * 0.693147182f + rl.y) + p;
}

// sin and cos of 2*pi*k24/2^24: the nearest 256th of a turn from the table, the rest (|b| <= pi/256) by
// sin b = b - b^3/6, cos b = 1 - b^2/2 and the addition theorems
template <typename Tab>
__device__ inline void sincos_turn(uint32_t k24, float& s, float& c, Tab tab)
{
    const uint32_t hi = (k24 + 0x8000u) >> 16;
    const int32_t lo = static_cast<int32_t>(k24) - static_cast<int32_t>(hi << 16);
    const float b = static_cast<float>(lo) * 3.74507039e-7f;
    const float z = b * b;
    const float sb = b + (b * z) * -0.166666672f;
    const float cb = 1.0f + z * -0.5f;
    const float2 a = tab[kRngLn + (hi & 255u)];
    s = a.x * cb + a.y * sb;
    c = a.y * cb - a.x * sb;
}

// Correctly rounded square root of t in [0, 2^6] (never subnormal unless 0): the hardware's 1-ulp estimate, moved to a
// neighbour when the residual says so -- the compiler's own sqrtf minus its scaling of tiny arguments and its
// zero / infinity patch (about half the instructions).  The result is the IEEE square root, i.e. what sqrtf returns on the CPU.
__device__ inline float sqrt_rn_small(float t)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const float y = __builtin_amdgcn_sqrtf(t);
    const float down = __uint_as_float(__float_as_uint(y) - 1u);      // t = 0: NaN, never taken
    const float up = __uint_as_float(__float_as_uint(y) + 1u);
    const float r_down = __builtin_fmaf(-down, y, t);
    const float r_up = __builtin_fmaf(-up, y, t);
    float r = (r_down <= 0.0f) ? down : y;
    r = (r_up > 0.0f) ? up : r;
    return r;
#else
    return sqrtf(t);
#endif
}

// Box-Muller on a 32-bit uniform u1 in [2^-32, 1] and a 24-bit angle: |z| <= sqrt(64 ln 2) sigma = 6.66 sigma
template <typename Tab>
__device__ inline float2 gaussian_pair(uint32_t x1, uint32_t x2, float sigma, Tab tab)
{
    const float u1 = (static_cast<float>(x1) + 1.0f) * 2.3283064365386963e-10f;
    float t = -2.0f * ln_unit(u1, tab);
    t = (t > 0.0f) ? t : 0.0f;                      // u1 = 1: the table form of ln may come out a few 1e-9 above 0
    const float rad = sqrt_rn_small(t);             // correctly rounded, t <= 64 ln 2
    float s, c;
    sincos_turn(x2 >> 8, s, c, tab);
    const float a = sigma * rad;
    return make_float2(a * c, a * s);
}

// Block shared by payload samples r and r + N/2.
__device__ inline Philox4 noise_block(uint64_t seed, uint64_t g, uint32_t r)
{
    return philox4x32_10(r, static_cast<uint32_t>(g), static_cast<uint32_t>(g >> 32), kStreamNoise,
                         static_cast<uint32_t>(seed), static_cast<uint32_t>(seed >> 32));
}
template <typename Tab>
__device__ inline float2 payload_noise(uint64_t seed, uint64_t g, uint32_t n, uint32_t n_fft, float sigma, Tab tab)
{
    const uint32_t half = n_fft >= 2 ? n_fft / 2 : 1;
    const Philox4 b = noise_block(seed, g, n % half);
    const bool hi = ((n / half) & 1u) != 0;
    return gaussian_pair(hi ? b.v[2] : b.v[0], hi ? b.v[3] : b.v[1], sigma, tab);
}
// Block shared by cyclic-prefix samples r and r + ceil(cp/2).
__device__ inline uint32_t cp_half(uint32_t cp) { return cp >= 2 ? (cp + 1) / 2 : 1; }
__device__ inline Philox4 cp_block(uint64_t seed, uint64_t g, uint32_t r)
{
    return philox4x32_10(r, static_cast<uint32_t>(g), static_cast<uint32_t>(g >> 32), kStreamCp,
                         static_cast<uint32_t>(seed), static_cast<uint32_t>(seed >> 32));
}
template <typename Tab>
__device__ inline float2 cp_noise(uint64_t seed, uint64_t g, uint32_t cidx, uint32_t cp, float sigma, Tab tab)
{
    const uint32_t half = cp_half(cp);
    const Philox4 b = cp_block(seed, g, cidx % half);
    const bool hi = ((cidx / half) & 1u) != 0;
    return gaussian_pair(hi ? b.v[2] : b.v[0], hi ? b.v[3] : b.v[1], sigma, tab);
}

// Channel coefficient of subcarrier k in set `set`: unit-variance complex Gaussian.
template <typename Tab>
__device__ inline float2 chan_sample(uint64_t seed, uint64_t set, uint32_t k, Tab tab)
{
    const Philox4 b = philox4x32_10(k, static_cast<uint32_t>(set), static_cast<uint32_t>(set >> 32), kStreamChan,
                                    static_cast<uint32_t>(seed), static_cast<uint32_t>(seed >> 32));
    return gaussian_pair(b.v[0], b.v[1], 0.70710678f, tab);
}

// Copies the tables into a workgroup's LDS (the caller barriers before the first use).
__device__ inline void rng_tables_to_lds(const float2* __restrict__ src, float2* lds_tab, int tid, int wg_threads)
{
    for (int i = tid; i < kRngPairs; i += wg_threads) lds_tab[i] = src[i];
}

} // namespace sdr
