// rng.hpp -- counter-based randomness of the device frame generator / AWGN channel.
// No reference counterpart.  Philox4x32-10 (Salmon et al., SC'11) plus
// elementary functions written with +,-,* only (and one correctly rounded sqrt), so that a CPU
// implementation of the same formulas yields identical bits (the build uses -ffp-contract=off).
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

// ln(x) for x in [2^-24, 1]
__device__ inline float ln_unit(float x)
{
    const uint32_t bits = __float_as_uint(x);
    int e = static_cast<int>(bits >> 23) - 127;
    float m = __uint_as_float((bits & 0x007FFFFFu) | 0x3F800000u);
    if (m > 1.41421356f) { m = m * 0.5f; e += 1; }
    const float f = m - 1.0f;
    const float z = f * f;
    float p = 7.0376836292e-2f;
    p = p * f + -1.1514610310e-1f;
    p = p * f + 1.1676998740e-1f;
    p = p * f + -1.2420140846e-1f;
    p = p * f + 1.4249322787e-1f;
    p = p * f + -1.6668057665e-1f;
    p = p * f + 2.0000714765e-1f;
    p = p * f + -2.4999993993e-1f;
    p = p * f + 3.3333331174e-1f;
    float y = (p * f) * z;
    const float fe = static_cast<float>(e);
    y = y + fe * -2.12194440e-4f;
    y = y + z * -0.5f;
    float r = f + y;
    r = r + fe * 0.693359375f;
    return r;
}

// sin and cos of 2*pi*k24/2^24
__device__ inline void sincos_turn(uint32_t k24, float& s, float& c)
{
    const uint32_t shifted = (k24 + 0x200000u) & 0xFFFFFFu;
    const uint32_t quad = shifted >> 22;
    const int32_t rem = static_cast<int32_t>(shifted & 0x3FFFFFu) - 0x200000;
    const float th = (static_cast<float>(rem) * 2.384185791015625e-07f) * 1.57079637f;
    const float z = th * th;
    float ps = 2.75573192e-6f;
    ps = ps * z + -1.98412698e-4f;
    ps = ps * z + 8.33333333e-3f;
    ps = ps * z + -1.66666667e-1f;
    const float sn = th + (th * z) * ps;
    float pc = 2.48015873e-5f;
    pc = pc * z + -1.38888889e-3f;
    pc = pc * z + 4.16666667e-2f;
    pc = pc * z + -0.5f;
    const float cs = 1.0f + z * pc;
    s = (quad == 0) ? sn : (quad == 1) ? cs : (quad == 2) ? -sn : -cs;
    c = (quad == 0) ? cs : (quad == 1) ? -sn : (quad == 2) ? -cs : sn;
}

__device__ inline float2 gaussian_pair(uint32_t x1, uint32_t x2, float sigma)
{
    const float u1 = static_cast<float>((x1 >> 8) + 1u) * 5.9604644775390625e-08f;
    const float rad = sqrtf(-2.0f * ln_unit(u1));   // correctly rounded (hipcc default); __fsqrt_rn is the native 1-ulp form
    float s, c;
    sincos_turn(x2 >> 8, s, c);
    const float a = sigma * rad;
    return make_float2(a * c, a * s);
}

// Block shared by payload samples r and r + N/2.
__device__ inline Philox4 noise_block(uint64_t seed, uint64_t g, uint32_t r)
{
    return philox4x32_10(r, static_cast<uint32_t>(g), static_cast<uint32_t>(g >> 32), kStreamNoise,
                         static_cast<uint32_t>(seed), static_cast<uint32_t>(seed >> 32));
}
__device__ inline float2 payload_noise(uint64_t seed, uint64_t g, uint32_t n, uint32_t n_fft, float sigma)
{
    const uint32_t half = n_fft >= 2 ? n_fft / 2 : 1;
    const Philox4 b = noise_block(seed, g, n % half);
    const bool hi = ((n / half) & 1u) != 0;
    return gaussian_pair(hi ? b.v[2] : b.v[0], hi ? b.v[3] : b.v[1], sigma);
}
__device__ inline float2 cp_noise(uint64_t seed, uint64_t g, uint32_t cidx, float sigma)
{
    const Philox4 b = philox4x32_10(cidx, static_cast<uint32_t>(g), static_cast<uint32_t>(g >> 32), kStreamCp,
                                    static_cast<uint32_t>(seed), static_cast<uint32_t>(seed >> 32));
    return gaussian_pair(b.v[0], b.v[1], sigma);
}

// Channel coefficient of subcarrier k in set `set`: unit-variance complex Gaussian.
__device__ inline float2 chan_sample(uint64_t seed, uint64_t set, uint32_t k)
{
    const Philox4 b = philox4x32_10(k, static_cast<uint32_t>(set), static_cast<uint32_t>(set >> 32), kStreamChan,
                                    static_cast<uint32_t>(seed), static_cast<uint32_t>(seed >> 32));
    return gaussian_pair(b.v[0], b.v[1], 0.70710678f);
}

} // namespace sdr
