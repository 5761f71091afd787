// qam.hpp -- device-side square-QAM arithmetic shared by the demapping kernels.
//
// Constellation rule (reference 16-QAM: lib/inc/modulation.hpp:15-47; other orders extend it):
// index bits MSB first, high MI bits -> I axis, low MQ bits -> Q axis, per-axis Gray code,
// level = 2*gray_decode(g) - (2^m - 1).  BPSK is MI = 1, MQ = 0 (Q level 0).
//
// nearest() (lib/inc/modulation.hpp:49-67) scans all 2^(MI+MQ) points in ascending index with
// d2 = dx*dx + dy*dy and a strict '<' from numeric_limits::max().  Here the same decision is
// obtained from the per-axis squared distances: rounding is monotonic, so the scan's minimum is
// m = fl(min_a DX[a] + min_b DY[b]), and the first index that attains it is the smallest I code
// gi with fl(DX[gi] + min_b DY[b]) == m followed by the smallest Q code gq with
// fl(DX[gi] + DY[gq]) == m.  Ties created by rounding (far-away points) therefore resolve to the
// lowest index exactly like the scan.  NaN / overflow (m not < max) leave index 0, also like it.
#pragma once

#include <hip/hip_runtime.h>

#ifndef SDR_FAST_FROM
#define SDR_FAST_FROM 2      // hard decision alone: bits per axis from which the margin slicer replaces the scan
#endif

namespace sdr {

template <typename T> struct FltMax;
template <> struct FltMax<float> { static __device__ __host__ constexpr float value() { return 3.402823466e+38f; } };
template <> struct FltMax<double> { static __device__ __host__ constexpr double value() { return 1.7976931348623157e+308; } };

__device__ __host__ constexpr unsigned gray_to_binary(unsigned g)
{
    return g ^ (g >> 1) ^ (g >> 2) ^ (g >> 3) ^ (g >> 4);   // enough for 5 bits per axis
}

template <typename T, int M>
__device__ __host__ constexpr T axis_level(unsigned g)
{
    return M == 0 ? T(0) : T(2 * static_cast<int>(gray_to_binary(g)) - ((1 << M) - 1));
}

template <typename T> __device__ inline T min_of(T a, T b) { return a < b ? a : b; }
template <> __device__ inline float min_of<float>(float a, float b) { return fminf(a, b); }
template <> __device__ inline double min_of<double>(double a, double b) { return fmin(a, b); }

// squared distances from coordinate v to every level of an axis with M bits
template <typename T, int M>
__device__ inline void axis_distances(T v, T (&d)[1 << M])
{
#pragma unroll
    for (unsigned g = 0; g < (1u << M); ++g) {
        const T diff = v - axis_level<T, M>(g);
        d[g] = diff * diff;
    }
}

template <typename T, int N>
__device__ inline T array_min(const T (&d)[N])
{
    T m = d[0];
#pragma unroll
    for (int i = 1; i < N; ++i) m = min_of<T>(m, d[i]);
    return m;
}

// Hard decision on the scaled point (ur, ui) = point * inorm, given the axis distances.
template <typename T, int MI, int MQ>
__device__ inline unsigned nearest_from_distances(const T (&dx)[1 << MI], const T (&dy)[1 << MQ])
{
    const T dxm = array_min<T, (1 << MI)>(dx);
    const T dym = array_min<T, (1 << MQ)>(dy);
    const T m = dxm + dym;
    unsigned gi = 0;
    T dxi = dx[0];
    bool hit = false;
#pragma unroll
    for (int g = (1 << MI) - 1; g >= 0; --g) {
        const bool eq = (dx[g] + dym) == m;
        gi = eq ? static_cast<unsigned>(g) : gi;
        dxi = eq ? dx[g] : dxi;
        hit = hit || eq;
    }
    unsigned gq = 0;
#pragma unroll
    for (int g = (1 << MQ) - 1; g >= 0; --g) {
        const bool eq = (dxi + dy[g]) == m;
        gq = eq ? static_cast<unsigned>(g) : gq;
    }
    const unsigned idx = (gi << MQ) | gq;
    return (hit && m < FltMax<T>::value()) ? idx : 0u;
}

// Max-log LLRs of one axis: for axis bit b (MSB first), min over levels with Gray bit 1 minus
// min over levels with Gray bit 0, times w.  out[0..M).
// d is indexed by the Gray code g, so "bit b of g is 1" is every second aligned block of 2^b entries:
// the minima of aligned blocks are shared between the bits (level s works on the block minima of
// size 2^s and hands the pairwise minima on to level s + 1): 82 instead of 150 min operations for
// 32 levels, 36 instead of 56 for 16.  min is exact and its NaN rule (the other operand) does not
// depend on the grouping, so the values are those of the plain double loop.
template <typename T, int M, int S>
__device__ inline void axis_llr_level(const T (&cur)[1 << (M - S)], T w, T* out)
{
    constexpr int CNT = 1 << (M - S);
    T d0 = cur[0], d1 = cur[1];
#pragma unroll
    for (int j = 2; j < CNT; j += 2) {
        d0 = min_of<T>(d0, cur[j]);
        d1 = min_of<T>(d1, cur[j + 1]);
    }
    out[M - 1 - S] = (d1 - d0) * w;
    if constexpr (S + 1 < M) {
        T next[CNT / 2];
#pragma unroll
        for (int j = 0; j < CNT / 2; ++j) next[j] = min_of<T>(cur[2 * j], cur[2 * j + 1]);
        axis_llr_level<T, M, S + 1>(next, w, out);
    }
}

template <typename T, int M>
__device__ inline void axis_llr(const T (&d)[1 << M], T w, T* out)
{
    if constexpr (M >= 5) {
        // 32 levels: 4979 instead of 5396 VALU instructions per wave and symbol at N = 1024; below
        // that the plain loops fold into three-input minima and are level (16) or shorter (8)
        axis_llr_level<T, M, 0>(d, w, out);
    } else {
#pragma unroll
        for (int b = M - 1; b >= 0; --b) {
            T d0 = T(0), d1 = T(0);
            bool h0 = false, h1 = false;
#pragma unroll
            for (unsigned g = 0; g < (1u << M); ++g) {
                if ((g >> b) & 1u) { d1 = h1 ? min_of<T>(d1, d[g]) : d[g]; h1 = true; }
                else               { d0 = h0 ? min_of<T>(d0, d[g]) : d[g]; h0 = true; }
            }
            out[M - 1 - b] = (d1 - d0) * w;
        }
    }
}

// The full evaluation, out of line: the fast path below falls back to it on (near-)ties, NaN and
// overflow, which random data meets about once in 10^7 points.
template <typename T, int MI, int MQ>
__device__ __noinline__ unsigned nearest_by_scan(T ur, T ui)
{
    T dx[1 << MI];
    T dy[1 << MQ];
    axis_distances<T, MI>(ur, dx);
    axis_distances<T, MQ>(ui, dy);
    return nearest_from_distances<T, MI, MQ>(dx, dy);
}

// One axis of the fast path: the slicer's level A (binary index), its squared distance and the
// smaller of the squared distances to its two neighbours (+inf beyond the edge of the grid).
template <typename T, int M>
__device__ inline void axis_candidate(T v, unsigned& code, T& d_best, T& d_next)
{
    if constexpr (M == 0) {
        code = 0u;
        d_best = v * v;
        d_next = FltMax<T>::value() * T(2);                       // +inf
    } else {
        constexpr int L = 1 << M;
        T a = rint((v + T(L - 1)) * T(0.5));
        a = a > T(0) ? a : T(0);                                  // also catches NaN
        a = a < T(L - 1) ? a : T(L - 1);
        const T lvl = a * T(2) - T(L - 1);
        const T dc = v - lvl, dl = v - (lvl - T(2)), dr = v - (lvl + T(2));
        d_best = dc * dc;
        const T inf = FltMax<T>::value() * T(2);
        const T left = a > T(0) ? dl * dl : inf;
        const T right = a < T(L - 1) ? dr * dr : inf;
        d_next = min_of<T>(left, right);
        const unsigned b = static_cast<unsigned>(static_cast<int>(a));
        code = b ^ (b >> 1);                                      // binary -> Gray
    }
}

// nearest() without the scan.  The slicer proposes (A, B); the proposal is the scan's answer
// whenever it beats, AFTER rounding, the best point that differs from it on either axis:
//   fl(next_I + d_B) > m  and  fl(d_A + next_Q) > m,   m = fl(d_A + d_B) < max.
// Proof sketch: the computed axis distances are monotonic in |v - level| (rounding is monotonic),
// so every other level is at least as far as the nearer neighbour; the second inequality forces
// d_B to be the strict minimum on Q and the first d_A on I, hence m is the scan's minimum and
// every other point rounds to something larger: the strict '<' scan from max() returns (A, B).
// Anything else (ties, absorbed differences, NaN, overflow) takes the scan itself.
template <typename T, int MI, int MQ>
__device__ inline unsigned nearest_fast(T ur, T ui)
{
    unsigned gi, gq;
    T da, na, db, nb;
    axis_candidate<T, MI>(ur, gi, da, na);
    axis_candidate<T, MQ>(ui, gq, db, nb);
    const T m = da + db;
    const bool sure = (na + db > m) && (da + nb > m) && (m < FltMax<T>::value());
    if (sure) return (gi << MQ) | gq;
    return nearest_by_scan<T, MI, MQ>(ur, ui);
}

__device__ __forceinline__ float fused_mul_add(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double fused_mul_add(double a, double b, double c) { return __builtin_fma(a, b, c); }

// One axis of the margin slicer: binary level index (as a float) and the distance of the slicer coordinate
//   t = fl(x * scale + (L - 1) / 2)        (one fused operation; slicer units: one unit = two grid units)
// from the centre of its decision cell.  (x, scale) is (re, inorm / 2) or (v, 1/2) with v = fl(re * inorm) the scan's
// own coordinate: x * scale is v / 2 up to the rounding of v, so t is within 2^-18 of v / 2 + (L - 1) / 2 while |t| < 64.
template <typename T, int M>
__device__ __forceinline__ void axis_slice(T x, T scale, T& level, T& miss)
{
    if constexpr (M == 0) {
        level = T(0);
        miss = T(0);
    } else {
        constexpr int L = 1 << M;
        const T t = fused_mul_add(x, scale, T(L - 1) * T(0.5));
        T tc = t > T(-0.25) ? t : T(-0.25);                       // NaN -> -0.25 (the reach test sees the NaN)
        tc = tc < T(L) - T(0.75) ? tc : T(L) - T(0.75);
        level = rint(tc);                                         // 0 ... L-1
        miss = tc - level;
    }
}
#ifndef SDR_MARGIN_OFF
#define SDR_MARGIN_OFF 0     // 1: experiment, every hard decision takes the three-point test
#endif

// nearest() from a plain slicer whenever the point is CLEARLY inside one decision cell, about a quarter
// of the instructions of the three-point test.  (A, B) is accepted when, on both axes, the slicer
// coordinate t is at most 1/2 - 2^-14 from the cell centre (the outer cells extend outwards) and the point's
// reach (|v_I| + |v_Q|) / 2 is at most R = (L - 1) / 2 + 31, tested as |x_I| + |x_Q| <= (R / scale) (1 - 2^-20) on the
// unscaled coordinates (a NaN on either axis fails this test: the sum, unlike a maximum, does not lose it).  Why this
// is the scan's answer, float or double:
//   * |t| <= R + (L - 1) / 2 < 64, so t is within 2^-18 of the slicer coordinate of v and v is at most 1 - 2^-13 + 2^-17
//     from level A and at least 1 + 2^-13 - 2^-17 from every other level of that axis; beyond the grid the
//     nearest other level is 2 further than the edge level;
//   * the scan's rounded axis distances are monotonic in |v - level|, so every other level's squared distance
//     exceeds A's by at least that of the neighbour: 4 (1 - 2^-4) 2^-13 > 1.8 * 2^-12 inside, more than 4 outside;
//   * the reach bound leaves a coordinate at most 30.5 slicer units beyond the outer cell of its axis (and the two
//     excesses together no more): the distance to the own cell's level is at most 62 grid units on one axis with at
//     most 1 + 2^-13 on the other (the sum of squares is convex, so the split excess is smaller), i.e. every sum m is
//     below 2^12, where one ulp is at most 2^-12: a point that differs from (A, B) on either axis has an exact sum
//     more than 1.8 ulp larger and therefore a strictly larger rounded one.  m is finite, so the strict '<' scan
//     returns (A, B).
// Everything else -- one point in 4000 of a uniform spread, NaN, far outliers -- takes the
// three-point test (and from there, rarely, the scan).
template <typename T, int MI, int MQ>
__device__ __noinline__ unsigned nearest_checked(T ur, T ui) { return nearest_fast<T, MI, MQ>(ur, ui); }

template <typename T, int MI, int MQ>
struct MarginSlicer {
    static constexpr int LMAX = 1 << (MI > MQ ? MI : MQ);
    // limit of |x_I| + |x_Q| for coordinates that still have to be multiplied by `scale` to become half coordinates
    static __device__ __forceinline__ T reach_limit(T scale) { return (T(LMAX - 1) * T(0.5) + T(31)) / scale * (T(1) - T(1.0 / 1048576.0)); }
    static __device__ __forceinline__ T miss_limit() { return T(0.5) - T(0.00006103515625); }
    // slicer proposal: both binary level indices as one number level_I * 2^MQ + level_Q (exact in T, not yet
    // Gray-coded), the larger distance from a cell centre (never NaN: the clamp) and the reach (NaN if either is)
    static __device__ __forceinline__ T propose(T xi, T xq, T scale, T& miss, T& reach)
    {
        T ai, aq, mi, mq;
        axis_slice<T, MI>(xi, scale, ai, mi);
        axis_slice<T, MQ>(xq, scale, aq, mq);
        miss = fmax(fabs(mi), fabs(mq));
        reach = fabs(xi) + fabs(xq);
        if constexpr (MQ == 0) return ai;
        else return fused_mul_add(ai, T(1 << MQ), aq);
    }
    static __device__ __forceinline__ bool sure(T miss, T reach, T limit) { return static_cast<bool>(static_cast<int>(miss <= miss_limit()) & static_cast<int>(reach <= limit)); }      // no short circuit: two compares and a scalar AND
    // Gray-codes both axes of the proposal in one go
    static __device__ __forceinline__ unsigned gray(unsigned w)
    {
        constexpr unsigned kKeep = MQ > 0 ? ~(1u << (MQ - 1)) : ~0u;   // I's lowest bit must not leak into Q's highest
        return w ^ ((w >> 1) & kKeep);
    }
};

// (xi, xq) * scale = half coordinates (exactly or to the last bit (ur, ui) / 2); (ur, ui) only feed the out-of-line test
template <typename T, int MI, int MQ>
__device__ inline unsigned nearest_margin(T ur, T ui)
{
    if constexpr (SDR_MARGIN_OFF) return nearest_fast<T, MI, MQ>(ur, ui);
    using MS = MarginSlicer<T, MI, MQ>;
    T miss, reach;
    const T w = MS::propose(ur, ui, T(0.5), miss, reach);
    if (MS::sure(miss, reach, MS::reach_limit(T(0.5)))) return MS::gray(static_cast<unsigned>(static_cast<int>(w)));
    return nearest_checked<T, MI, MQ>(ur, ui);
}

// LLR magnitude statistic (counter 5, no reference counterpart): every LLR adds
//   q(L) = round(min(|L|, 64) / 64 * 65535)        (NaN adds 0; a tie cannot occur: x * 65535 is never k + 1/2)
// i.e. |L| clipped at 64 in steps of 64/65535 = 0.977e-3 -- an integer, so that sums do not depend on the order
// of reduction.  This is what v_cvt_pknorm_u16_f32 computes for two values at once (checked against the formula over
// every float in [0, 1] on MI355X: tools/pknormcheck.hip), and v_dot2_u32_u16 adds both halves to a 32-bit sum:
// two instructions per pair of LLRs plus the exact scaling by 2^-6, where trunc(min(|L| * 1024, 2^30)) of rounds 1-2
// cost four per LLR.  `acc` must have room for 2 * 65535 more.
typedef unsigned short sdr_u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned llr_abs_add2(float l0, float l1, unsigned acc)
{
    const sdr_u16x2 q = __builtin_amdgcn_cvt_pknorm_u16(__builtin_fabsf(l0 * 0.015625f), __builtin_fabsf(l1 * 0.015625f));
    const sdr_u16x2 ones = {1, 1};
    return __builtin_amdgcn_udot2(q, ones, acc, false);
}
// sum over `COUNT` LLRs held in an array
template <int COUNT>
__device__ __forceinline__ unsigned llr_abs_sum(const float* l, unsigned acc)
{
#pragma unroll
    for (int i = 0; i + 1 < COUNT; i += 2) acc = llr_abs_add2(l[i], l[i + 1], acc);
    if constexpr (COUNT % 2 == 1) acc = llr_abs_add2(l[COUNT - 1], 0.0f, acc);
    return acc;
}

// Everything for one subcarrier: scaled coordinates -> LLRs (MI+MQ values) and index.
template <typename T, int MI, int MQ, bool WANT_LLR, bool WANT_IDX>
__device__ inline unsigned demap_point(T re, T im, T inorm, T w, T* llr)
{
    if constexpr (!WANT_LLR) {
        // hard decision alone: no need for every distance (4-level axes gain nothing from it); the slicer works on the
        // unscaled point (one fused operation per axis), the scan's own coordinates are formed only where the
        // out-of-line test wants them
        if constexpr (WANT_IDX && (MI >= SDR_FAST_FROM || MQ >= SDR_FAST_FROM)) {
            if constexpr (SDR_MARGIN_OFF) return nearest_fast<T, MI, MQ>(re * inorm, im * inorm);
            using MS = MarginSlicer<T, MI, MQ>;
            const T half = inorm * T(0.5);
            T miss, reach;
            const T w2 = MS::propose(re, im, half, miss, reach);
            if (MS::sure(miss, reach, MS::reach_limit(half))) return MS::gray(static_cast<unsigned>(static_cast<int>(w2)));
            return nearest_checked<T, MI, MQ>(re * inorm, im * inorm);
        }
    }
    const T ur = re * inorm;
    const T ui = im * inorm;
    T dx[1 << MI];
    T dy[1 << MQ];
    axis_distances<T, MI>(ur, dx);
    axis_distances<T, MQ>(ui, dy);
    if (WANT_LLR) {
        axis_llr<T, MI>(dx, w, llr);
        if (MQ > 0) axis_llr<T, MQ>(dy, w, llr + MI);
    }
    unsigned idx = 0;
    if constexpr (WANT_IDX) {
        // even with every distance at hand the three-point test is cheaper than the scan from 8 levels up
        if constexpr (MI >= SDR_FAST_FROM || MQ >= SDR_FAST_FROM) idx = nearest_margin<T, MI, MQ>(ur, ui);
        else idx = nearest_from_distances<T, MI, MQ>(dx, dy);
    }
    return idx;
}

// Runtime-modulation dispatch helper: calls f.template operator()<MI,MQ>() for the given bps.
template <typename F>
__device__ __host__ inline bool dispatch_mod(int bps, F&& f)
{
    switch (bps) {
        case 1: f.template operator()<1, 0>(); return true;
        case 2: f.template operator()<1, 1>(); return true;
        case 4: f.template operator()<2, 2>(); return true;
        case 6: f.template operator()<3, 3>(); return true;
        case 8: f.template operator()<4, 4>(); return true;
        case 10: f.template operator()<5, 5>(); return true;
        default: return false;
    }
}

inline bool valid_mod(int bps) { return bps == 1 || bps == 2 || bps == 4 || bps == 6 || bps == 8 || bps == 10; }

// Host: Es on the integer grid, and the reference's norm / inorm taken in T
// (lib/inc/modulation.hpp:20,23: norm = T{1}/sqrt(T{10}), inorm = sqrt(T{10})).
inline double grid_energy(int bps)
{
    if (bps == 1) return 1.0;
    const double l = static_cast<double>(1 << (bps / 2));
    return 2.0 * (l * l - 1.0) / 3.0;
}

} // namespace sdr
