// sdr_oracle.cpp -- CPU restatement of the erithion/SDR receive-side hot path.
//
// TEST INFRASTRUCTURE ONLY.  Nothing under sdr_amd/ or include/ may include, link or
// call this file; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do,
// and there only as the checker / the timed CPU baseline.
//
// Pinning status: the reference (header-only C++23, needs <expected>, which libstdc++ 11 in
// this image lacks) is unbuildable here without a stand-in header, so it is NOT compiled.
// This restatement is pinned by the reference's own tests for the path
// (lib/tests/fft_test.cpp:10-72, lib/tests/ofdm_test.cpp:11-36), by the reference output
// recorded in SURVEY.md 8(c) (fft2 of 0..7 in float), and cross-checked against numpy.fft.
// Everything marked "X*" below (equaliser, LLR, QPSK/64/256-QAM, AWGN, generator, BER) has no
// counterpart in the reference: parity UNPINNED there; this file is the definition.
//
// Build: g++ -O2 -ffp-contract=off (see oracle/Makefile).  No -ffast-math, no FMA
// contraction: every floating-point operation below rounds once, in the order written.
// (The reference built by its own CMake for baseline x86-64 has no FMA either.)
//
// All citations are path:line relative to /root/reference.

#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <limits>
#include <thread>
#include <vector>

namespace {

const char kPow2Error[] = "The sequence size must be of powers of 2";   // lib/inc/fft.hpp:92

// ---------------------------------------------------------------------------------------------
// R3  bit-reversal permutation                                   lib/inc/fft.hpp:35-51
// Walks i upward while keeping j = bit-reversed(i) by a reversed-carry increment, and
// exchanges the two elements once per unordered pair.
// ---------------------------------------------------------------------------------------------
template <typename T>
void bit_reverse_inplace(T* x /* interleaved re,im */, std::size_t size)
{
    std::size_t j = 0;
    for (std::size_t i = 1; i < size; ++i) {
        std::size_t bit = size >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;   // clear the run of ones from the top
        j |= bit;                              // set the first zero: reversed "+1"
        if (i < j) {
            T re = x[2 * i], im = x[2 * i + 1];
            x[2 * i] = x[2 * j];       x[2 * i + 1] = x[2 * j + 1];
            x[2 * j] = re;             x[2 * j + 1] = im;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// R2  radix-2 DIT Cooley-Tukey, iterative, in place            lib/inc/fft.hpp:87-155
// Per stage (span = 2,4,..,size): theta is formed in double and narrowed to T (:114),
// cos/sin are taken in T (:115-116).  Per block the twiddle starts at 1+0i (:122-123); per
// butterfly: t = odd*w written out as four products (:130-134), a' = even+t, b' = even-t
// (:136-137); then w *= (cos,sin) (:140-142) and, whenever (j & 31)==0 (j==0 included), w is
// renormalised by hypot (:145-150).
// Returns 0 on success, 1 when size is neither 0 nor a power of two (:91-92).
// ---------------------------------------------------------------------------------------------
template <typename T>
int radix2_dit(T* x, std::size_t size, bool inverse)
{
    if (size > 0 && (size & (size - 1)) != 0) return 1;
    bit_reverse_inplace(x, size);
    for (std::size_t span = 2; span <= size; span <<= 1) {
        const T theta = static_cast<T>(-2.0 * (inverse ? -1.0 : 1.0) * 3.141592653589793238462643383279502884 / static_cast<double>(span));
        const T c = std::cos(theta);
        const T s = std::sin(theta);
        const std::size_t half = span / 2;
        for (std::size_t blk = 0; blk < size; blk += span) {
            T wr = static_cast<T>(1.0);
            T wi = static_cast<T>(0.0);
            for (std::size_t j = 0; j < half; ++j) {
                T* lo = x + 2 * (blk + j);
                T* hi = x + 2 * (blk + j + half);
                const T er = lo[0], ei = lo[1];
                const T pr = hi[0], pi = hi[1];
                const T tr = pr * wr - pi * wi;
                const T ti = pr * wi + pi * wr;
                lo[0] = er + tr;  lo[1] = ei + ti;
                hi[0] = er - tr;  hi[1] = ei - ti;
                const T keep = wr;
                wr = keep * c - wi * s;
                wi = keep * s + wi * c;
                if ((j & 31) == 0) {
                    const T mag = std::hypot(wr, wi);
                    wr /= mag;
                    wi /= mag;
                }
            }
        }
    }
    return 0;
}

// The same twiddle sequence as a table: entry (span/2 - 1 + j) holds the w used by butterfly
// j of the stage with that span.  The blocks of one stage all restart at 1+0i, so the
// sequence is block independent.  Used by tests to check the library's
// own table generator bit for bit.
template <typename T>
void twiddle_table(std::size_t size, bool inverse, T* out /* 2*(size-1) */)
{
    for (std::size_t span = 2; span <= size; span <<= 1) {
        const T theta = static_cast<T>(-2.0 * (inverse ? -1.0 : 1.0) * 3.141592653589793238462643383279502884 / static_cast<double>(span));
        const T c = std::cos(theta);
        const T s = std::sin(theta);
        const std::size_t half = span / 2;
        T wr = static_cast<T>(1.0), wi = static_cast<T>(0.0);
        for (std::size_t j = 0; j < half; ++j) {
            out[2 * (half - 1 + j)] = wr;
            out[2 * (half - 1 + j) + 1] = wi;
            const T keep = wr;
            wr = keep * c - wi * s;
            wi = keep * s + wi * c;
            if ((j & 31) == 0) {
                const T mag = std::hypot(wr, wi);
                wr /= mag;
                wi /= mag;
            }
        }
    }
}

// R4  inverse = same network with +theta, then every element divided by the length converted
// to T (complex /= real is two real divisions)                  lib/inc/fft.hpp:194-205
template <typename T>
int radix2_dit_inverse_scaled(T* x, std::size_t size)
{
    const int rc = radix2_dit(x, size, true);
    if (rc) return rc;
    const T len = static_cast<T>(static_cast<std::ptrdiff_t>(size));
    for (std::size_t i = 0; i < 2 * size; ++i) x[i] /= len;
    return 0;
}

// R1  ofdm::rx: drop the first cp samples, transform the rest   lib/inc/ofdm.hpp:37-43
// (in.size() < cp is an unguarded size_t underflow in the reference; here rc=2.)
template <typename T>
int ofdm_rx(const T* in, std::size_t n_in, std::size_t cp, T* out)
{
    if (n_in < cp) return 2;
    const std::size_t n = n_in - cp;
    std::memcpy(out, in + 2 * cp, 2 * n * sizeof(T));
    return radix2_dit(out, n, false);
}

// R5  ofdm::tx: payload after a cp-sized gap, inverse transform in place, then the last cp
// samples are copied to the front                               lib/inc/ofdm.hpp:13-24
template <typename T>
int ofdm_tx(const T* in, std::size_t n, std::size_t cp, T* out)
{
    std::memcpy(out + 2 * cp, in, 2 * n * sizeof(T));
    const int rc = radix2_dit_inverse_scaled(out + 2 * cp, n);
    if (rc) return rc;
    // std::copy(out.end()-cp, out.end(), out.begin()) walks forward (ofdm.hpp:21); the
    // source index n+i is always ahead of the destination i, so it reads untouched samples.
    for (std::size_t i = 0; i < 2 * cp; ++i) out[i] = out[2 * n + i];
    return 0;
}

// ---------------------------------------------------------------------------------------------
// Constellations.
// R6: 16-QAM of the reference                                   lib/inc/modulation.hpp:15-68
//   index = b3b2b1b0, I level from b3b2, Q level from b1b0, per-axis Gray map
//   00->-3 01->-1 11->+1 10->+3 (table :29-47), norm = 1/sqrt(10) (:20), inorm = sqrt(10) (:23).
// X3 (not in the reference, parity unpinned): the same rule for any square QAM with
//   bps = 2m bits: high m bits -> I, low m bits -> Q, level = 2*gray_decode(g) - (2^m - 1),
//   inorm = sqrt(2(4^m-1)/3) taken in T, norm = 1/inorm taken in T.  bps = 1 (BPSK) is the
//   one-axis case: index 0 -> -1, 1 -> +1, Q level 0, inorm = norm = 1.
// ---------------------------------------------------------------------------------------------
inline unsigned gray_decode(unsigned g)
{
    unsigned b = 0;
    for (; g; g >>= 1) b ^= g;
    return b;
}

struct Constellation {
    int bps = 0;
    int m_i = 0, m_q = 0;          // bits on the I and on the Q axis
    unsigned points = 0;
    double energy = 1.0;           // mean |point|^2 on the integer grid
};

inline bool make_constellation(int bps, Constellation& c)
{
    if (!(bps == 1 || bps == 2 || bps == 4 || bps == 6 || bps == 8 || bps == 10)) return false;
    c.bps = bps;
    c.m_i = (bps == 1) ? 1 : bps / 2;
    c.m_q = (bps == 1) ? 0 : bps / 2;
    c.points = 1u << bps;
    const double li = static_cast<double>(1u << c.m_i);
    c.energy = (bps == 1) ? 1.0 : 2.0 * (li * li - 1.0) / 3.0;
    return true;
}

template <typename T> inline T axis_level(unsigned gray, int m)
{
    if (m == 0) return static_cast<T>(0);
    return static_cast<T>(2 * static_cast<int>(gray_decode(gray)) - ((1 << m) - 1));
}
template <typename T> inline T level_i(const Constellation& c, unsigned idx) { return axis_level<T>(idx >> c.m_q, c.m_i); }
template <typename T> inline T level_q(const Constellation& c, unsigned idx) { return axis_level<T>(idx & ((1u << c.m_q) - 1u), c.m_q); }
template <typename T> inline T inorm_of(const Constellation& c) { return std::sqrt(static_cast<T>(c.energy)); }
template <typename T> inline T norm_of(const Constellation& c) { return static_cast<T>(1) / std::sqrt(static_cast<T>(c.energy)); }

// R6 nearest: scale the point by inorm (complex*real = two products, :54), then scan the
// table in ascending index with a strict '<' against a running best that starts at
// numeric_limits<T>::max() (:52-65).  Lowest index wins ties; NaN or overflowing input
// leaves index 0.  X3 uses the identical scan over its own table.
template <typename T>
unsigned nearest_index(const Constellation& c, T re, T im)
{
    const T scale = inorm_of<T>(c);
    const T ur = re * scale;
    const T ui = im * scale;
    unsigned res = 0;
    T best = std::numeric_limits<T>::max();
    for (unsigned i = 0; i != c.points; ++i) {
        const T dx = ur - level_i<T>(c, i);
        const T dy = ui - level_q<T>(c, i);
        const T d2 = dx * dx + dy * dy;
        if (d2 < best) { best = d2; res = i; }
    }
    return res;
}

// MSB-first bit stream helpers: symbol k occupies stream bits [k*bps, (k+1)*bps).
// For bps = 4 this is exactly "high nibble first" of to_constl/from_constl
// (lib/inc/modulation.hpp:87-91, :105-108).
inline unsigned stream_get(const std::uint8_t* bytes, std::size_t bitpos, int nbits)
{
    unsigned v = 0;
    for (int b = 0; b < nbits; ++b, ++bitpos)
        v = (v << 1) | ((bytes[bitpos >> 3] >> (7 - (bitpos & 7))) & 1u);
    return v;
}
inline void stream_put(std::uint8_t* bytes, std::size_t bitpos, int nbits, unsigned v)
{
    for (int b = nbits - 1; b >= 0; --b, ++bitpos)
        if ((v >> b) & 1u) bytes[bitpos >> 3] |= static_cast<std::uint8_t>(0x80u >> (bitpos & 7));
}

// Writes the leading bits of v that fall before limit_bits (a symbol straddling the end of the
// last whole byte contributes only those).
inline void stream_put_clipped(std::uint8_t* bytes, std::size_t bitpos, int nbits, unsigned v, std::size_t limit_bits)
{
    if (bitpos >= limit_bits) return;
    if (bitpos + nbits <= limit_bits) { stream_put(bytes, bitpos, nbits, v); return; }
    const int keep = static_cast<int>(limit_bits - bitpos);
    stream_put(bytes, bitpos, keep, v >> (nbits - keep));
}

// R8 to_constl: each index becomes table[index]*norm, a complex*real product (:90-91).
template <typename T>
void map_indices(const Constellation& c, const std::uint8_t* bytes, std::size_t n_sym, T* out)
{
    const T nrm = norm_of<T>(c);
    for (std::size_t k = 0; k < n_sym; ++k) {
        const unsigned idx = stream_get(bytes, k * c.bps, c.bps);
        out[2 * k] = level_i<T>(c, idx) * nrm;
        out[2 * k + 1] = level_q<T>(c, idx) * nrm;
    }
}

// R7 from_constl: hard decisions packed MSB first; symbols that do not complete a byte are
// dropped (for 16-QAM: the trailing unpaired symbol, :103).  Returns the byte count.
template <typename T>
std::size_t demap_hard(const Constellation& c, const T* sym, std::size_t n_sym, std::uint8_t* out)
{
    const std::size_t n_bytes = (n_sym * c.bps) / 8;
    std::memset(out, 0, n_bytes);
    for (std::size_t k = 0; k < n_sym && k * c.bps < n_bytes * 8; ++k)
        stream_put_clipped(out, k * c.bps, c.bps, nearest_index<T>(c, sym[2 * k], sym[2 * k + 1]), n_bytes * 8);
    return n_bytes;
}

// ---------------------------------------------------------------------------------------------
// X1 one-tap equaliser (not in the reference): yhat = y*G as four products,
//    re = yr*gr - yi*gi, im = yr*gi + yi*gr.
// X2 max-log LLR (not in the reference), on the un-normalised grid like nearest (:54):
//    u = yhat*inorm; for the bit at position b of the index (b = 0 is the MSB; I-axis bits
//    first, then Q-axis bits) and v the matching coordinate of u:
//       D0 = min over axis levels whose Gray bit is 0 of (v-l)*(v-l), D1 likewise for bit 1,
//       LLR = (D1 - D0) * w,      w = norm^2 * |H_k|^2 / N0 supplied by the caller.
//    Positive LLR <=> bit 0 more likely.  This is the exact max-log metric for square Gray
//    QAM (the other axis cancels).  Hard bits come from nearest_index, never from LLR signs.
// ---------------------------------------------------------------------------------------------
template <typename T>
void soft_demap_point(const Constellation& c, T re, T im, T w, T* llr /* bps */)
{
    const T scale = inorm_of<T>(c);
    const T u[2] = { re * scale, im * scale };
    const int mbits[2] = { c.m_i, c.m_q };
    int o = 0;
    for (int axis = 0; axis < 2; ++axis) {
        const int m = mbits[axis];
        for (int b = m - 1; b >= 0; --b) {           // MSB of the axis Gray code first
            T d[2] = { std::numeric_limits<T>::infinity(), std::numeric_limits<T>::infinity() };
            for (unsigned g = 0; g < (1u << m); ++g) {
                const T diff = u[axis] - axis_level<T>(g, m);
                const T sq = diff * diff;
                const unsigned bit = (g >> b) & 1u;
                if (sq < d[bit]) d[bit] = sq;
            }
            llr[o++] = (d[1] - d[0]) * w;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// X4 counter-based randomness (not in the reference).  Philox4x32-10 (Salmon et al., SC'11).
// ---------------------------------------------------------------------------------------------
inline void philox4x32_10(const std::uint32_t ctr[4], const std::uint32_t key[2], std::uint32_t out[4])
{
    std::uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    std::uint32_t k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; ++r) {
        const std::uint64_t p0 = static_cast<std::uint64_t>(0xD2511F53u) * c0;
        const std::uint64_t p1 = static_cast<std::uint64_t>(0xCD9E8D57u) * c2;
        const std::uint32_t n0 = static_cast<std::uint32_t>(p1 >> 32) ^ c1 ^ k0;
        const std::uint32_t n1 = static_cast<std::uint32_t>(p1);
        const std::uint32_t n2 = static_cast<std::uint32_t>(p0 >> 32) ^ c3 ^ k1;
        const std::uint32_t n3 = static_cast<std::uint32_t>(p0);
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

constexpr std::uint32_t kStreamTx = 0x54580000u;      // transmitted indices
constexpr std::uint32_t kStreamNoise = 0x4E5A0000u;   // payload noise
constexpr std::uint32_t kStreamCp = 0x43500000u;      // noise on cyclic-prefix samples
constexpr std::uint32_t kStreamChan = 0x43480000u;    // channel realisations (one per set of symbols)

// Transmitted constellation index of subcarrier k of global OFDM symbol g:
// one Philox block serves the 8 subcarriers {r, r+stride, ...}, stride = max(N/8,1);
// subcarrier k takes 16-bit lane k/stride of block k%stride, masked to bps bits.
inline unsigned tx_index(std::uint64_t seed, std::uint64_t g, std::uint32_t k, std::uint32_t n_fft, int bps)
{
    const std::uint32_t stride = n_fft >= 8 ? n_fft / 8 : 1;
    const std::uint32_t ctr[4] = { k % stride, static_cast<std::uint32_t>(g), static_cast<std::uint32_t>(g >> 32), kStreamTx };
    const std::uint32_t key[2] = { static_cast<std::uint32_t>(seed), static_cast<std::uint32_t>(seed >> 32) };
    std::uint32_t r[4];
    philox4x32_10(ctr, key, r);
    const std::uint32_t lane = (k / stride) & 7u;
    const std::uint32_t v = (r[lane >> 1] >> (16 * (lane & 1))) & 0xFFFFu;
    return v & ((1u << bps) - 1u);
}

// Own elementary functions so that CPU and GPU produce identical bits: only +,-,* on floats,
// integer bit manipulation, one correctly rounded sqrt and two small tables that both sides build with
// the same double-precision formulas (round 3: table-based range reduction, about a third of the
// operations of the long polynomials of rounds 1-2).
//   ln table, 32 entries: interval i of the mantissa m in [1,2) has centre c_i = 1 + (i + 1/2)/32;
//     entry = (r_i, L_i), r_i = fl(1/c_i), L_i = fl(-ln r_i); then ln m = L_i + log1p(f), f = m r_i - 1, |f| < 1/63
//   sin/cos table, 256 entries: (sin, cos)(2 pi j / 256); the angle 2 pi k / 2^24 is j / 256 of a turn plus b, |b| <= pi/256
struct RngTables {
    float ln_r[32], ln_l[32], sn[256], cs[256];
    RngTables()
    {
        for (int i = 0; i < 32; ++i) {
            const double c = 1.0 + (static_cast<double>(i) + 0.5) / 32.0;
            ln_r[i] = static_cast<float>(1.0 / c);
            ln_l[i] = static_cast<float>(-std::log(static_cast<double>(ln_r[i])));
        }
        for (int j = 0; j < 256; ++j) {
            const double a = 2.0 * 3.14159265358979323846264338327950288 * static_cast<double>(j) / 256.0;
            sn[j] = static_cast<float>(std::sin(a));
            cs[j] = static_cast<float>(std::cos(a));
        }
    }
};
inline const RngTables& rng_tables() { static const RngTables t; return t; }

inline float ln_unit(float x)   // x in [2^-32, 1], normal
{
    const RngTables& tab = rng_tables();
    std::uint32_t bits;
    std::memcpy(&bits, &x, 4);
    const int e = static_cast<int>(bits >> 23) - 127;
    const std::uint32_t idx = (bits >> 18) & 31u;
    const std::uint32_t mb = (bits & 0x007FFFFFu) | 0x3F800000u;
    float m;
    std::memcpy(&m, &mb, 4);                    // [1,2)
    const float f = m * tab.ln_r[idx] - 1.0f;
    const float q = f * 0.333333343f + -0.5f;
    const float z = f * f;
    const float p = z * q + f;                  // log1p(f) = f - f^2/2 + f^3/3, |error| < 2e-8
    const float fe = static_cast<float>(e);
    return (fe * 0.693147182f + tab.ln_l[idx]) + p;
}

inline void sincos_turn(std::uint32_t k24, float& s, float& c)   // angle = 2*pi*k24/2^24
{
    const RngTables& tab = rng_tables();
    const std::uint32_t hi = (k24 + 0x8000u) >> 16;                          // nearest 256th of a turn, 0..256
    const std::int32_t lo = static_cast<std::int32_t>(k24) - static_cast<std::int32_t>(hi << 16);   // -32768 .. 32767
    const float b = static_cast<float>(lo) * 3.74507039e-7f;                 // 2 pi / 2^24
    const float z = b * b;
    const float sb = b + (b * z) * -0.166666672f;
    const float cb = 1.0f + z * -0.5f;
    const float sa = tab.sn[hi & 255u], ca = tab.cs[hi & 255u];
    s = sa * cb + ca * sb;
    c = ca * cb - sa * sb;
}

// Box-Muller on a 32-bit uniform u1 in [2^-32, 1] and a 24-bit angle: |z| <= sqrt(64 ln 2) sigma = 6.66 sigma
// (a true Gaussian holds 2.7e-11 beyond that, per component).  Rounds 1-2 used 24 bits for u1 (5.77 sigma, 8e-9).
inline void gaussian_pair(std::uint32_t x1, std::uint32_t x2, float sigma, float& re, float& im)
{
    const float u1 = (static_cast<float>(x1) + 1.0f) * 2.3283064365386963e-10f;   // (0,1]
    float t = -2.0f * ln_unit(u1);
    if (!(t > 0.0f)) t = 0.0f;                   // u1 = 1: the table form of ln may come out a few 1e-9 above 0
    const float rad = std::sqrt(t);
    float s, c;
    sincos_turn(x2 >> 8, s, c);
    const float a = sigma * rad;
    re = a * c;
    im = a * s;
}

// Noise on payload sample n (0..N-1, CP excluded) of global symbol g: samples n and n+N/2
// share one Philox block; sigma is the per-component standard deviation.
inline void payload_noise(std::uint64_t seed, std::uint64_t g, std::uint32_t n, std::uint32_t n_fft, float sigma, float& re, float& im)
{
    const std::uint32_t half = n_fft >= 2 ? n_fft / 2 : 1;
    const std::uint32_t ctr[4] = { n % half, static_cast<std::uint32_t>(g), static_cast<std::uint32_t>(g >> 32), kStreamNoise };
    const std::uint32_t key[2] = { static_cast<std::uint32_t>(seed), static_cast<std::uint32_t>(seed >> 32) };
    std::uint32_t r[4];
    philox4x32_10(ctr, key, r);
    const std::uint32_t h = (n / half) & 1u;
    gaussian_pair(r[2 * h], r[2 * h + 1], sigma, re, im);
}
// Noise on cyclic-prefix sample cidx (0..cp-1): samples c and c + ceil(cp/2) share one Philox block.
inline void cp_noise(std::uint64_t seed, std::uint64_t g, std::uint32_t cidx, std::uint32_t cp, float sigma, float& re, float& im)
{
    const std::uint32_t half = cp >= 2 ? (cp + 1) / 2 : 1;
    const std::uint32_t ctr[4] = { cidx % half, static_cast<std::uint32_t>(g), static_cast<std::uint32_t>(g >> 32), kStreamCp };
    const std::uint32_t key[2] = { static_cast<std::uint32_t>(seed), static_cast<std::uint32_t>(seed >> 32) };
    std::uint32_t r[4];
    philox4x32_10(ctr, key, r);
    const std::uint32_t h = (cidx / half) & 1u;
    gaussian_pair(r[2 * h], r[2 * h + 1], sigma, re, im);
}

// One transmitted+received OFDM symbol (model: app/QFDemoWindow.cpp:201-224 with a channel
// inserted): indices -> to_constl (R8) -> optional per-subcarrier channel H (four-product
// complex multiply) -> tx (R5) -> + AWGN.  rx_out has N+cp samples.
void generate_symbol(const Constellation& c, std::uint32_t n_fft, std::uint32_t cp, std::uint64_t seed, std::uint64_t g,
                     float sigma, const float* chan /* N c64 or null */, float* rx_out, unsigned* idx_out /* N or null */)
{
    std::vector<float> freq(2 * static_cast<std::size_t>(n_fft));
    const float nrm = norm_of<float>(c);
    for (std::uint32_t k = 0; k < n_fft; ++k) {
        const unsigned idx = tx_index(seed, g, k, n_fft, c.bps);
        if (idx_out) idx_out[k] = idx;
        const float xr = level_i<float>(c, idx) * nrm;
        const float xi = level_q<float>(c, idx) * nrm;
        if (chan) {
            const float hr = chan[2 * k], hi = chan[2 * k + 1];
            freq[2 * k] = xr * hr - xi * hi;
            freq[2 * k + 1] = xr * hi + xi * hr;
        } else {
            freq[2 * k] = xr;
            freq[2 * k + 1] = xi;
        }
    }
    ofdm_tx<float>(freq.data(), n_fft, cp, rx_out);
    for (std::uint32_t i = 0; i < cp; ++i) {
        float nr, ni;
        cp_noise(seed, g, i, cp, sigma, nr, ni);
        rx_out[2 * i] += nr;
        rx_out[2 * i + 1] += ni;
    }
    for (std::uint32_t n = 0; n < n_fft; ++n) {
        float nr, ni;
        payload_noise(seed, g, n, n_fft, sigma, nr, ni);
        rx_out[2 * (cp + n)] += nr;
        rx_out[2 * (cp + n) + 1] += ni;
    }
}

// Receive one symbol: rx (R1), equalise (X1), LLR (X2), nearest (R6/X3).
void demod_symbol(const Constellation& c, std::uint32_t n_fft, std::uint32_t cp, const float* rx_in,
                  const float* eq /* N c64 or null */, const float* wgt /* N or null */, float w0,
                  float* fft_out /* N c64 or null */, float* llr_out /* N*bps or null */, unsigned* idx_out /* N or null */,
                  bool w_from_taps = false /* no weight array: w_k = w0 / |G_k|^2 (SDR_DEMOD_WEIGHT_FROM_TAPS) */)
{
    std::vector<float> y(2 * static_cast<std::size_t>(n_fft));
    ofdm_rx<float>(rx_in, static_cast<std::size_t>(n_fft) + cp, cp, y.data());
    if (fft_out) std::memcpy(fft_out, y.data(), y.size() * sizeof(float));
    for (std::uint32_t k = 0; k < n_fft; ++k) {
        float yr = y[2 * k], yi = y[2 * k + 1];
        float w = wgt ? wgt[k] : w0;
        if (eq) {
            const float gr = eq[2 * k], gi = eq[2 * k + 1];
            const float er = yr * gr - yi * gi;
            const float ei = yr * gi + yi * gr;
            yr = er; yi = ei;
            if (w_from_taps && !wgt) w = w0 / (gr * gr + gi * gi);
        }
        if (llr_out) soft_demap_point<float>(c, yr, yi, w, llr_out + static_cast<std::size_t>(k) * c.bps);
        if (idx_out) idx_out[k] = nearest_index<float>(c, yr, yi);
    }
}

inline std::uint64_t popcount32(std::uint32_t v) { return static_cast<std::uint64_t>(__builtin_popcount(v)); }

// LLR statistics kept as integers so that sums do not depend on reduction order:
//   q = round(min(|L|, 64) / 64 * 65535) (|L| clipped at 64, steps of 64/65535; NaN counts 0; the product is exact in
//   double and never lies on a tie), and a sign error is (L < 0) != transmitted bit.
inline void llr_stats(const float* llr, int bps, unsigned tx_idx, std::uint64_t& abs_q10, std::uint64_t& sign_err)
{
    for (int b = 0; b < bps; ++b) {
        const float l = llr[b];
        const float x = std::fabs(l) * 0.015625f;        // exact (a power of two)
        if (x == x) abs_q10 += static_cast<std::uint64_t>(std::floor(static_cast<double>(x < 1.0f ? x : 1.0f) * 65535.0 + 0.5));
        const unsigned txbit = (tx_idx >> (bps - 1 - b)) & 1u;
        const unsigned rxbit = (l < 0.0f) ? 1u : 0u;
        sign_err += (txbit != rxbit);
    }
}

} // namespace

// =============================================================================================
// C entry points (ctypes).  Complex buffers are interleaved re,im.
// =============================================================================================
extern "C" {

const char* orc_pow2_error(void) { return kPow2Error; }

int orc_fft2_c32(float* x, std::size_t n) { return radix2_dit<float>(x, n, false); }
int orc_fft2_c64(double* x, std::size_t n) { return radix2_dit<double>(x, n, false); }
int orc_ifft2_c32(float* x, std::size_t n) { return radix2_dit_inverse_scaled<float>(x, n); }
int orc_ifft2_c64(double* x, std::size_t n) { return radix2_dit_inverse_scaled<double>(x, n); }
int orc_fft2_batch_c32(float* x, std::size_t n, std::size_t batch, int inverse)
{
    for (std::size_t b = 0; b < batch; ++b) {
        const int rc = inverse ? radix2_dit_inverse_scaled<float>(x + 2 * b * n, n) : radix2_dit<float>(x + 2 * b * n, n, false);
        if (rc) return rc;
    }
    return 0;
}
int orc_fft2_batch_c64(double* x, std::size_t n, std::size_t batch, int inverse)
{
    for (std::size_t b = 0; b < batch; ++b) {
        const int rc = inverse ? radix2_dit_inverse_scaled<double>(x + 2 * b * n, n) : radix2_dit<double>(x + 2 * b * n, n, false);
        if (rc) return rc;
    }
    return 0;
}
void orc_bit_reverse_c32(float* x, std::size_t n) { bit_reverse_inplace<float>(x, n); }
void orc_twiddles_c32(std::size_t n, int inverse, float* out) { twiddle_table<float>(n, inverse != 0, out); }
void orc_twiddles_c64(std::size_t n, int inverse, double* out) { twiddle_table<double>(n, inverse != 0, out); }

int orc_ofdm_rx_c32(const float* in, std::size_t n_in, std::size_t cp, float* out) { return ofdm_rx<float>(in, n_in, cp, out); }
int orc_ofdm_rx_c64(const double* in, std::size_t n_in, std::size_t cp, double* out) { return ofdm_rx<double>(in, n_in, cp, out); }
int orc_ofdm_tx_c32(const float* in, std::size_t n, std::size_t cp, float* out) { return ofdm_tx<float>(in, n, cp, out); }
int orc_ofdm_tx_c64(const double* in, std::size_t n, std::size_t cp, double* out) { return ofdm_tx<double>(in, n, cp, out); }
int orc_ofdm_rx_batch_c32(const float* in, std::size_t n_fft, std::size_t cp, std::size_t n_sym, float* out)
{
    for (std::size_t s = 0; s < n_sym; ++s) {
        const int rc = ofdm_rx<float>(in + 2 * s * (n_fft + cp), n_fft + cp, cp, out + 2 * s * n_fft);
        if (rc) return rc;
    }
    return 0;
}
int orc_ofdm_tx_batch_c32(const float* in, std::size_t n_fft, std::size_t cp, std::size_t n_sym, float* out)
{
    for (std::size_t s = 0; s < n_sym; ++s) {
        const int rc = ofdm_tx<float>(in + 2 * s * n_fft, n_fft, cp, out + 2 * s * (n_fft + cp));
        if (rc) return rc;
    }
    return 0;
}

// constellation description: returns 0 on success
int orc_constellation_f32(int bps, float* norm, float* inorm, float* table /* 2*2^bps */)
{
    Constellation c;
    if (!make_constellation(bps, c)) return 1;
    *norm = norm_of<float>(c);
    *inorm = inorm_of<float>(c);
    for (unsigned i = 0; i < c.points; ++i) { table[2 * i] = level_i<float>(c, i); table[2 * i + 1] = level_q<float>(c, i); }
    return 0;
}
int orc_constellation_f64(int bps, double* norm, double* inorm, double* table)
{
    Constellation c;
    if (!make_constellation(bps, c)) return 1;
    *norm = norm_of<double>(c);
    *inorm = inorm_of<double>(c);
    for (unsigned i = 0; i < c.points; ++i) { table[2 * i] = level_i<double>(c, i); table[2 * i + 1] = level_q<double>(c, i); }
    return 0;
}
int orc_nearest_f32(int bps, const float* sym, std::size_t n, std::uint16_t* out)
{
    Constellation c;
    if (!make_constellation(bps, c)) return 1;
    for (std::size_t k = 0; k < n; ++k) out[k] = static_cast<std::uint16_t>(nearest_index<float>(c, sym[2 * k], sym[2 * k + 1]));
    return 0;
}
int orc_nearest_f64(int bps, const double* sym, std::size_t n, std::uint16_t* out)
{
    Constellation c;
    if (!make_constellation(bps, c)) return 1;
    for (std::size_t k = 0; k < n; ++k) out[k] = static_cast<std::uint16_t>(nearest_index<double>(c, sym[2 * k], sym[2 * k + 1]));
    return 0;
}
int orc_to_constl_f32(int bps, const std::uint8_t* bytes, std::size_t n_sym, float* out)
{
    Constellation c;
    if (!make_constellation(bps, c)) return 1;
    map_indices<float>(c, bytes, n_sym, out);
    return 0;
}
int orc_to_constl_f64(int bps, const std::uint8_t* bytes, std::size_t n_sym, double* out)
{
    Constellation c;
    if (!make_constellation(bps, c)) return 1;
    map_indices<double>(c, bytes, n_sym, out);
    return 0;
}
long long orc_from_constl_f32(int bps, const float* sym, std::size_t n_sym, std::uint8_t* out)
{
    Constellation c;
    if (!make_constellation(bps, c)) return -1;
    return static_cast<long long>(demap_hard<float>(c, sym, n_sym, out));
}
long long orc_from_constl_f64(int bps, const double* sym, std::size_t n_sym, std::uint8_t* out)
{
    Constellation c;
    if (!make_constellation(bps, c)) return -1;
    return static_cast<long long>(demap_hard<double>(c, sym, n_sym, out));
}
int orc_soft_demap_f32(int bps, const float* sym, std::size_t n_sym, const float* wgt, float w0, float* llr)
{
    Constellation c;
    if (!make_constellation(bps, c)) return 1;
    for (std::size_t k = 0; k < n_sym; ++k)
        soft_demap_point<float>(c, sym[2 * k], sym[2 * k + 1], wgt ? wgt[k] : w0, llr + k * c.bps);
    return 0;
}

// Zero-forcing taps and LLR weights from a channel estimate (X1/X2 helpers):
//   p = hr*hr + hi*hi;  G = (hr/p, -hi/p);  w = (norm*norm) * p * inv_n0
int orc_zf_taps_f32(int bps, const float* chan, std::size_t n, float inv_n0, float* taps, float* wgt)
{
    Constellation c;
    if (!make_constellation(bps, c)) return 1;
    const float nrm = norm_of<float>(c);
    const float n2 = nrm * nrm;
    for (std::size_t k = 0; k < n; ++k) {
        const float hr = chan[2 * k], hi = chan[2 * k + 1];
        const float p = hr * hr + hi * hi;
        taps[2 * k] = hr / p;
        taps[2 * k + 1] = -hi / p;
        wgt[k] = (n2 * p) * inv_n0;
    }
    return 0;
}

// Fused receive path over a batch (R1 + X1 + X2 + R6/X3).  Set s of the per-subcarrier
// arrays applies to symbols [s*sym_per_set, (s+1)*sym_per_set); sym_per_set = 0 means one set.
// hard_out receives floor(n_sym*N*bps/8) bytes, MSB first over the whole batch.
int orc_demod_flags_c32(std::uint32_t n_fft, std::uint32_t cp, int bps, std::size_t n_sym, std::uint32_t sym_per_set,
                        const float* rx, const float* eq, const float* wgt, float w0,
                        float* fft_out, float* llr_out, std::uint8_t* hard_out, unsigned flags /* 1: weights from the taps */)
{
    Constellation c;
    if (!make_constellation(bps, c)) return 1;
    if (n_fft > 0 && (n_fft & (n_fft - 1)) != 0) return 1;
    std::vector<unsigned> idx(n_fft);
    const std::size_t total_bits = n_sym * n_fft * static_cast<std::size_t>(bps);
    if (hard_out) std::memset(hard_out, 0, total_bits / 8);
    for (std::size_t s = 0; s < n_sym; ++s) {
        const std::size_t set = sym_per_set ? s / sym_per_set : 0;
        demod_symbol(c, n_fft, cp, rx + 2 * s * (static_cast<std::size_t>(n_fft) + cp),
                     eq ? eq + 2 * set * n_fft : nullptr, wgt ? wgt + set * n_fft : nullptr, w0,
                     fft_out ? fft_out + 2 * s * n_fft : nullptr,
                     llr_out ? llr_out + s * n_fft * bps : nullptr,
                     hard_out ? idx.data() : nullptr, (flags & 1u) != 0);
        if (hard_out)
            for (std::uint32_t k = 0; k < n_fft; ++k) {
                const std::size_t pos = (s * n_fft + k) * static_cast<std::size_t>(bps);
                stream_put_clipped(hard_out, pos, bps, idx[k], (total_bits / 8) * 8);
            }
    }
    return 0;
}

int orc_demod_c32(std::uint32_t n_fft, std::uint32_t cp, int bps, std::size_t n_sym, std::uint32_t sym_per_set,
                  const float* rx, const float* eq, const float* wgt, float w0,
                  float* fft_out, float* llr_out, std::uint8_t* hard_out)
{
    return orc_demod_flags_c32(n_fft, cp, bps, n_sym, sym_per_set, rx, eq, wgt, w0, fft_out, llr_out, hard_out, 0u);
}

// The same over n_threads host threads, symbols split into contiguous ranges (symbols are
// independent, lib/inc/ofdm.hpp:37-43 keeps no state; SURVEY.md 8(d) CPU baseline (ii)).  The
// reference itself is single-threaded (its OpenMP pragma, fft.hpp:119, is dead).  Needs
// n_fft*bps to be a whole number of bytes so that the ranges' hard bits do not share a byte.
int orc_demod_c32_threads(unsigned n_threads, std::uint32_t n_fft, std::uint32_t cp, int bps, std::size_t n_sym, std::uint32_t sym_per_set,
                          const float* rx, const float* eq, const float* wgt, float w0,
                          float* fft_out, float* llr_out, std::uint8_t* hard_out, unsigned flags)
{
    if (n_threads <= 1 || n_sym < 2 * n_threads) return orc_demod_flags_c32(n_fft, cp, bps, n_sym, sym_per_set, rx, eq, wgt, w0, fft_out, llr_out, hard_out, flags);
    const std::size_t sym_bits = static_cast<std::size_t>(n_fft) * static_cast<std::size_t>(bps);
    if (sym_bits % 8 != 0) return 1;
    const std::size_t unit = sym_per_set ? sym_per_set : 1;          // ranges start on a set boundary
    const std::size_t units = (n_sym + unit - 1) / unit;
    std::vector<std::thread> pool;
    std::vector<int> rc(n_threads, 0);
    for (unsigned t = 0; t < n_threads; ++t) {
        const std::size_t first = std::min(n_sym, (units * t / n_threads) * unit);
        const std::size_t last = std::min(n_sym, (units * (t + 1) / n_threads) * unit);
        if (first >= last) continue;
        pool.emplace_back([=, &rc] {
            const std::size_t set0 = sym_per_set ? first / sym_per_set : 0;
            rc[t] = orc_demod_flags_c32(n_fft, cp, bps, last - first, sym_per_set,
                                  rx + 2 * first * (static_cast<std::size_t>(n_fft) + cp),
                                  eq ? eq + 2 * set0 * n_fft : nullptr, wgt ? wgt + set0 * n_fft : nullptr, w0,
                                  fft_out ? fft_out + 2 * first * n_fft : nullptr,
                                  llr_out ? llr_out + first * sym_bits : nullptr,
                                  hard_out ? hard_out + first * sym_bits / 8 : nullptr, flags);
        });
    }
    for (auto& th : pool) th.join();
    for (int r : rc) if (r) return r;
    return 0;
}

void orc_philox4x32_10(const std::uint32_t* ctr, const std::uint32_t* key, std::uint32_t* out) { philox4x32_10(ctr, key, out); }
unsigned orc_tx_index(std::uint64_t seed, std::uint64_t g, std::uint32_t k, std::uint32_t n_fft, int bps) { return tx_index(seed, g, k, n_fft, bps); }
void orc_payload_noise(std::uint64_t seed, std::uint64_t g, std::uint32_t n, std::uint32_t n_fft, float sigma, float* out2)
{
    payload_noise(seed, g, n, n_fft, sigma, out2[0], out2[1]);
}
void orc_cp_noise(std::uint64_t seed, std::uint64_t g, std::uint32_t c, std::uint32_t cp, float sigma, float* out2) { cp_noise(seed, g, c, cp, sigma, out2[0], out2[1]); }
void orc_gaussian_pair(std::uint32_t x1, std::uint32_t x2, float sigma, float* out2) { gaussian_pair(x1, x2, sigma, out2[0], out2[1]); }
void orc_rng_tables(float* out /* 64 + 512: (r_i, L_i) pairs, then (sin, cos) pairs */)
{
    const RngTables& t = rng_tables();
    for (int i = 0; i < 32; ++i) { out[2 * i] = t.ln_r[i]; out[2 * i + 1] = t.ln_l[i]; }
    for (int j = 0; j < 256; ++j) { out[64 + 2 * j] = t.sn[j]; out[64 + 2 * j + 1] = t.cs[j]; }
}
float orc_ln_unit(float x) { return ln_unit(x); }
void orc_sincos_turn(std::uint32_t k24, float* out2) { sincos_turn(k24, out2[0], out2[1]); }

// Frame generator (X4): symbols g0 .. g0+n_sym-1.  chan: optional per-set channel H
// (n_sets*N c64; set of symbol g is (g / sym_per_set) - first_set with first_set = g0/sym_per_set).
// tx_bits: optional packed transmitted bits (MSB first over the batch).
int orc_link_generate_c32(std::uint32_t n_fft, std::uint32_t cp, int bps, std::uint64_t seed, std::uint64_t g0, std::size_t n_sym,
                          std::uint32_t sym_per_set, float sigma, const float* chan, float* rx_out, std::uint8_t* tx_bits)
{
    Constellation c;
    if (!make_constellation(bps, c)) return 1;
    if (n_fft == 0 || (n_fft & (n_fft - 1)) != 0) return 1;
    std::vector<unsigned> idx(n_fft);
    const std::size_t total_bits = n_sym * n_fft * static_cast<std::size_t>(bps);
    if (tx_bits) std::memset(tx_bits, 0, total_bits / 8);
    const std::uint64_t first_set = sym_per_set ? g0 / sym_per_set : 0;
    for (std::size_t s = 0; s < n_sym; ++s) {
        const std::uint64_t g = g0 + s;
        const std::size_t set = sym_per_set ? static_cast<std::size_t>(g / sym_per_set - first_set) : 0;
        generate_symbol(c, n_fft, cp, seed, g, sigma, chan ? chan + 2 * set * n_fft : nullptr,
                        rx_out + 2 * s * (static_cast<std::size_t>(n_fft) + cp), idx.data());
        if (tx_bits)
            for (std::uint32_t k = 0; k < n_fft; ++k) {
                const std::size_t pos = (s * n_fft + k) * static_cast<std::size_t>(bps);
                stream_put_clipped(tx_bits, pos, bps, idx[k], (total_bits / 8) * 8);
            }
    }
    return 0;
}

// Monte-Carlo link (X4): generate -> demodulate -> count, nothing stored.
// counters[0] bit errors, [1] bits, [2] subcarrier-symbol errors, [3] subcarrier symbols,
// [4] OFDM symbols, [5] sum of round(min(|LLR|, 64) / 64 * 65535), [6] LLR sign errors, [7] reserved.
int orc_link_ber_c32(std::uint32_t n_fft, std::uint32_t cp, int bps, std::uint64_t seed, std::uint64_t g0, std::size_t n_sym,
                     float sigma, float w0, std::uint64_t* counters /* 8, accumulated into */)
{
    Constellation c;
    if (!make_constellation(bps, c)) return 1;
    if (n_fft == 0 || (n_fft & (n_fft - 1)) != 0) return 1;
    std::vector<unsigned> tx(n_fft), rx(n_fft);
    std::vector<float> samples(2 * (static_cast<std::size_t>(n_fft) + cp));
    std::vector<float> llr(static_cast<std::size_t>(n_fft) * bps);
    for (std::size_t s = 0; s < n_sym; ++s) {
        const std::uint64_t g = g0 + s;
        generate_symbol(c, n_fft, cp, seed, g, sigma, nullptr, samples.data(), tx.data());
        demod_symbol(c, n_fft, cp, samples.data(), nullptr, nullptr, w0, nullptr, llr.data(), rx.data());
        for (std::uint32_t k = 0; k < n_fft; ++k) {
            const std::uint32_t diff = tx[k] ^ rx[k];
            counters[0] += popcount32(diff);
            counters[2] += (diff != 0);
            llr_stats(llr.data() + static_cast<std::size_t>(k) * bps, bps, tx[k], counters[5], counters[6]);
        }
        counters[1] += static_cast<std::uint64_t>(n_fft) * bps;
        counters[3] += n_fft;
        counters[4] += 1;
    }
    return 0;
}

// Channel realisations (X4, BASELINE.json configs[2]: "random unit-variance complex Gaussian H_k, one
// set per 14-symbol frame"): H[set][k] = Box-Muller pair of Philox block (k, set, stream tag) with
// standard deviation sqrt(1/2) per component.  Sets first_set .. first_set + n_sets - 1.
int orc_link_channel_c32(std::uint32_t n_fft, std::uint64_t seed, std::uint64_t first_set, std::size_t n_sets, float* chan_out)
{
    const std::uint32_t key[2] = { static_cast<std::uint32_t>(seed), static_cast<std::uint32_t>(seed >> 32) };
    for (std::size_t s = 0; s < n_sets; ++s) {
        const std::uint64_t set = first_set + s;
        for (std::uint32_t k = 0; k < n_fft; ++k) {
            const std::uint32_t ctr[4] = { k, static_cast<std::uint32_t>(set), static_cast<std::uint32_t>(set >> 32), kStreamChan };
            std::uint32_t r[4];
            philox4x32_10(ctr, key, r);
            gaussian_pair(r[0], r[1], 0.70710678f, chan_out[2 * (s * n_fft + k)], chan_out[2 * (s * n_fft + k) + 1]);
        }
    }
    return 0;
}

// The whole faded link (configs[2]): per set of sym_per_set symbols a channel realisation, zero-forcing
// taps and LLR weights from it (perfect channel knowledge), generate -> demodulate with them -> count.
// g0 must be a multiple of sym_per_set.  llr_out (optional): n_sym*N*bps LLRs.
int orc_link_chain_c32(std::uint32_t n_fft, std::uint32_t cp, int bps, std::uint64_t seed, std::uint64_t g0, std::size_t n_sym, std::uint32_t sym_per_set,
                       float sigma, float inv_n0, std::uint64_t* counters /* 8, accumulated into */, float* llr_out)
{
    Constellation c;
    if (!make_constellation(bps, c)) return 1;
    if (n_fft == 0 || (n_fft & (n_fft - 1)) != 0 || sym_per_set == 0 || g0 % sym_per_set != 0) return 1;
    std::vector<unsigned> tx(n_fft), rx(n_fft);
    std::vector<float> samples(2 * (static_cast<std::size_t>(n_fft) + cp));
    std::vector<float> llr(static_cast<std::size_t>(n_fft) * bps);
    std::vector<float> chan(2 * static_cast<std::size_t>(n_fft)), taps(2 * static_cast<std::size_t>(n_fft)), wgt(n_fft);
    std::uint64_t have_set = ~std::uint64_t(0);
    for (std::size_t s = 0; s < n_sym; ++s) {
        const std::uint64_t g = g0 + s;
        const std::uint64_t set = g / sym_per_set;
        if (set != have_set) {
            orc_link_channel_c32(n_fft, seed, set, 1, chan.data());
            orc_zf_taps_f32(bps, chan.data(), n_fft, inv_n0, taps.data(), wgt.data());
            have_set = set;
        }
        generate_symbol(c, n_fft, cp, seed, g, sigma, chan.data(), samples.data(), tx.data());
        demod_symbol(c, n_fft, cp, samples.data(), taps.data(), wgt.data(), 0.0f, nullptr, llr.data(), rx.data());
        if (llr_out) std::memcpy(llr_out + s * llr.size(), llr.data(), llr.size() * sizeof(float));
        for (std::uint32_t k = 0; k < n_fft; ++k) {
            const std::uint32_t diff = tx[k] ^ rx[k];
            counters[0] += popcount32(diff);
            counters[2] += (diff != 0);
            llr_stats(llr.data() + static_cast<std::size_t>(k) * bps, bps, tx[k], counters[5], counters[6]);
        }
        counters[1] += static_cast<std::uint64_t>(n_fft) * bps;
        counters[3] += n_fft;
        counters[4] += 1;
    }
    return 0;
}

// Bit errors between two packed streams.
std::uint64_t orc_count_bit_errors(const std::uint8_t* a, const std::uint8_t* b, std::size_t n_bytes)
{
    std::uint64_t e = 0;
    for (std::size_t i = 0; i < n_bytes; ++i) e += popcount32(static_cast<std::uint32_t>(a[i] ^ b[i]));
    return e;
}

} // extern "C"
