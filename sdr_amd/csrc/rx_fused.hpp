// rx_fused.hpp -- the tuned fused receive kernel (fp32, n_fft = 64..8192):
//   CP removal -> FFT (reference radix-2 DIT network) -> one-tap equalise -> max-log LLR
//   (+ optional nearest() hard bits, FFT bins, BER/LLR counters).
//
// Replaces, for whole batches of OFDM symbols, the reference call chain
//   ofdm::rx (lib/inc/ofdm.hpp:37-43) -> fft::fft2 (lib/inc/fft.hpp:87-155)
//   -> modulation::from_constl / nearest (lib/inc/modulation.hpp:49-112)
// with the equaliser and LLR steps (no reference counterpart) inserted between them.
//
// Arithmetic: exactly the reference's operation DAG.  After the bit reversal (lib/inc/fft.hpp:
// 35-51, here a free index remap on the global load) stage c pairs positions differing in bit c
// and multiplies the upper one by the table twiddle W_c[pos mod 2^c] as four products
// (:130-134), then adds/subtracts (:136-137).  The plan (fft_plans.hpp) only decides where the
// positions live: each thread keeps E of them in registers and runs 3-4 stages per pass with no
// data movement; between passes the array crosses LDS once in a bank-conflict-free layout.
// Built with -ffp-contract=off, so every product and sum rounds separately as on the CPU.
//
// Mapping: TPS = N/E threads per symbol; a workgroup is max(64, TPS) threads = 64/TPS symbols
// (small N) or one symbol (N >= 1024) and handles ONE symbol group as straight-line code (the
// tap-holding variants: two).  Twiddles: first pass = lane-uniform (scalar registers), middle pass =
// small LDS table, last pass = per-thread registers loaded once per workgroup.
// HBM traffic per symbol: 8*N bytes read (CP never fetched) + 4*N*bps bytes of LLR written.
//
// Instantiations <LOG2N, MI, MQ, FULL, EQ, HARDONLY>: MI/MQ = bits on the I/Q axis (0,0 = FFT
// only); FULL = hard bits / bins / counters optional at run time; EQ = per-subcarrier taps / weights
// held in registers; HARDONLY = packed hard decisions as the only output.  RxShape holds each
// instantiation's LDS layout, control structure and waves-per-SIMD bounds; its per-shape rules were
// each adopted from same-buffer A/Bs of two builds (tools/abbench.py; numbers in profiles/r2_*.txt and
// DESIGN.md section 6.1).  The compile-time experiment switches of rounds 1-2 that lost or measured
// level (shared taps through LDS, non-temporal tap loads, plain LLR stores, taps per symbol at N = 8192,
// ...) were removed in round 3; their results stay in profiles/.  Run-time knobs: see RxTuning.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <utility>

#include "bitpack.hpp"
#include "fft_plans.hpp"
#include "qam.hpp"

namespace sdr {

struct RxParams {
    const float2* rx;            // n_sym x (N + cp)
    const float2* tw;            // N-1 forward twiddles (reference recurrence)
    const float2* eq;            // n_sets x N or null
    const float* wgt;            // n_sets x N or null
    float* llr;                  // n_sym x N x bps or null
    uint8_t* hard;               // packed bits or null                 (FULL only)
    float2* fft;                 // n_sym x N or null                   (FULL / FFT-only)
    const uint8_t* ref;          // reference packed bits or null       (FULL only)
    unsigned long long* counters;// the context's counter slots, or null     (FULL only)
    unsigned long long n_sym;
    uint32_t cp;
    uint32_t sym_per_set;
    float w0;
    float inorm;
    uint32_t w_from_taps;        // no weight array: w_k = w0 / |G_k|^2 instead of the scalar w0 (SDR_DEMOD_WEIGHT_FROM_TAPS)
    uint32_t groups_per_wg;      // consecutive symbol groups handled by one workgroup
    uint32_t xcd_window;         // workgroup-id permutation window (multiple of 8; 0 = none)
    uint32_t xcd_stagger;        // XCD x starts x * xcd_stagger workgroups into its share of the window (wrapping round)
};

template <int LOG2N>
struct PlanOps {
    using P = FftPlan<LOG2N>;
    static constexpr int kN = 1 << LOG2N;
    static constexpr int kLogE = P::kLogE;
    static constexpr int kE = 1 << kLogE;
    static constexpr int kNT = P::kThreadBits;
    static constexpr int kTPS = 1 << kNT;                       // threads per symbol
    static constexpr int kWG = kTPS < 64 ? 64 : kTPS;            // workgroup size
    static constexpr int kSPW = kWG / kTPS;                      // symbols per workgroup
    static constexpr int kPasses = P::kPasses;
    static constexpr int kLast = kPasses - 1;

    static constexpr int reg_pos(int pass, int m)
    {
        int p = 0;
        for (int i = 0; i < kLogE; ++i)
            if ((m >> i) & 1) p |= 1 << P::kRegBit[pass][i];
        return p;
    }
    static constexpr int reg_addr(int ex, int pass, int m)
    {
        int a = 0;
        for (int i = 0; i < kLogE; ++i)
            if ((m >> i) & 1) a += P::kWeight[ex][P::kRegBit[pass][i]];
        return a;
    }
    static constexpr int bitrev(int v, int bits)
    {
        int r = 0;
        for (int i = 0; i < bits; ++i)
            if ((v >> i) & 1) r |= 1 << (bits - 1 - i);
        return r;
    }
    // register-bit index that carries stage c in this pass
    static constexpr int bfly_bit(int pass, int c)
    {
        for (int i = 0; i < kLogE; ++i)
            if (P::kRegBit[pass][i] == c) return i;
        return -1;
    }
    static constexpr int insert_zero(int h, int bit)
    {
        const int lo = h & ((1 << bit) - 1);
        return ((h >> bit) << (bit + 1)) | lo;
    }
    // number of group registers that are adjacent in the output (last pass) / input (first pass)
    static constexpr int kGroupOut = 1 << (kLogE - P::kR[kLast]);
    static constexpr int kGroupIn = 1 << (kLogE - P::kR[0]);
    static constexpr int max_sym_bytes()
    {
        int m = 0;
        for (int k = 0; k < kPasses - 1; ++k) m = P::kSymBytes[k] > m ? P::kSymBytes[k] : m;
        return (m + 15) & ~15;
    }
    // middle-pass twiddle table in LDS: entries of stages c0..c0+r-1 of pass 1 (3-pass plans only)
    static constexpr int kMidFirst = kPasses == 3 ? (1 << P::kC0[1]) - 1 : 0;
    static constexpr int kMidCount = kPasses == 3 ? (1 << (P::kC0[1] + P::kR[1])) - (1 << P::kC0[1]) : 0;
    // last-pass register twiddles: per stage, one per distinct low-register pattern
    static constexpr int last_low_mask(int c)   // register-bit indices whose position bit is below c
    {
        int mask = 0;
        for (int i = 0; i < kLogE; ++i)
            if (P::kRegBit[kLast][i] < c) mask |= 1 << i;
        return mask;
    }
    static constexpr int popcount(int v) { int c = 0; for (; v; v &= v - 1) ++c; return c; }
    static constexpr int pext(int v, int mask)
    {
        int r = 0, o = 0;
        for (int i = 0; i < kLogE; ++i)
            if ((mask >> i) & 1) { if ((v >> i) & 1) r |= 1 << o; ++o; }
        return r;
    }
    static constexpr int pdep(int v, int mask)
    {
        int r = 0, o = 0;
        for (int i = 0; i < kLogE; ++i)
            if ((mask >> i) & 1) { if ((v >> o) & 1) r |= 1 << i; ++o; }
        return r;
    }
    static constexpr int last_tw_offset(int s)   // slot offset of stage s (0-based inside the last pass)
    {
        int off = 0;
        for (int q = 0; q < s; ++q) off += 1 << popcount(last_low_mask(P::kC0[kLast] + q));
        return off;
    }
    static constexpr int kLastTwCount = last_tw_offset(P::kR[kLast]);

    template <int PASS>
    static __device__ __forceinline__ int thr_pos(int t)
    {
        int p = 0;
        static_for<kNT>([&](auto U) { constexpr int u = U; constexpr int pb = P::kThrBit[PASS][u]; p |= ((t >> u) & 1) << pb; });
        return p;
    }
    template <int EX, int PASS>
    static __device__ __forceinline__ int thr_addr(int t)
    {
        int a = 0;
        static_for<kNT>([&](auto U) { constexpr int u = U; constexpr int wgt = P::kWeight[EX][P::kThrBit[PASS][u]]; a += ((t >> u) & 1) * wgt; });
        return a;
    }
};

// CONJ: rotate by conj(w).  a - b*(-c) and a + b*c are the same IEEE operation, and the inverse
// table is the conjugate of the forward one except for the sign of the zero imaginary part of
// its W = 1 entries (checked on the host tables), so an inverse transform run this way on the
// forward table can differ from the true one only in the sign of exact zeros.
// UNIT: the twiddle is the recurrence's start value (1, +0) (lib/inc/fft.hpp:122-123; butterfly j = 0 of every block),
// known at compile time in the first pass.  x * 1 is x, so only the products with the zero remain -- they are kept:
// they carry the sign of the zero and turn an infinity or NaN into the NaN the reference gets.
template <bool CONJ = false, bool UNIT = false>
__device__ __forceinline__ void bfly(float2& even, float2& odd, const float2 w)
{
    float tr, ti;
    if constexpr (UNIT && CONJ) {
        tr = odd.x + odd.y * 0.0f;
        ti = odd.y - odd.x * 0.0f;
    } else if constexpr (UNIT) {
        tr = odd.x - odd.y * 0.0f;
        ti = odd.x * 0.0f + odd.y;
    } else if constexpr (CONJ) {
        tr = odd.x * w.x + odd.y * w.y;
        ti = odd.y * w.x - odd.x * w.y;
    } else {
        tr = odd.x * w.x - odd.y * w.y;
        ti = odd.x * w.y + odd.y * w.x;
    }
    const float ar = even.x + tr, ai = even.y + ti;
    const float br = even.x - tr, bi = even.y - ti;
    even.x = ar; even.y = ai;
    odd.x = br; odd.y = bi;
}

// stores COUNT consecutive floats with the widest access the chunk alignment allows
template <int COUNT>
__device__ __forceinline__ void store_floats(float* __restrict__ dst, const float* v)
{
    if constexpr (COUNT % 4 == 0) {
        static_for<COUNT / 4>([&](auto I) { constexpr int i = I; reinterpret_cast<float4*>(dst)[i] = make_float4(v[4 * i], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3]); });
    } else if constexpr (COUNT % 2 == 0) {
        static_for<COUNT / 2>([&](auto I) { constexpr int i = I; reinterpret_cast<float2*>(dst)[i] = make_float2(v[2 * i], v[2 * i + 1]); });
    } else {
        static_for<COUNT>([&](auto I) { constexpr int i = I; dst[i] = v[i]; });
    }
}


// Per-thread state of one plan: LDS addresses of both exchanges and the index parts contributed
// by the thread bits.  Shared by the forward and the inverse transform (same network).
template <int LOG2N>
struct FftAddr {
    using O = PlanOps<LOG2N>;
    using P = typename O::P;
    int wr0, rd0, wr1, rd1;      // byte offsets into the symbol's exchange region
    int mid_jt;                  // twiddle index part of the middle pass
    int pos_last;                // working-array position bits held by this thread in the last pass (= output bin bits)
    int in_thread;               // input index bits held by this thread in the first pass

    __device__ __forceinline__ void init(int t, int slot, int sym_bytes)
    {
        constexpr int LAST = O::kLast;
        pos_last = O::template thr_pos<LAST>(t);
        in_thread = 0;   // bitrev_n(thr_pos<0>(t))
        static_for<O::kNT>([&](auto U) { constexpr int u = U; constexpr int ib = LOG2N - 1 - P::kThrBit[0][u]; in_thread |= ((t >> u) & 1) << ib; });
        wr0 = slot * sym_bytes + O::template thr_addr<0, 0>(t);
        rd0 = slot * sym_bytes + O::template thr_addr<0, 1>(t);
        wr1 = rd1 = mid_jt = 0;
        if constexpr (O::kPasses == 3) {
            wr1 = slot * sym_bytes + O::template thr_addr<1, 1>(t);
            rd1 = slot * sym_bytes + O::template thr_addr<1, 2>(t);
            mid_jt = O::template thr_pos<1>(t) & ((1 << P::kC0[1]) - 1);
        }
    }
    // input index (first pass) / output index (last pass) of register m
    static constexpr int in_reg(int m) { return O::bitrev(O::reg_pos(0, m), LOG2N); }
    static constexpr int out_reg(int m) { return O::reg_pos(O::kLast, m); }
};

// Twiddles of one direction: first pass lane-uniform (scalar registers), middle pass in an LDS
// table, last pass in per-thread registers.
template <int LOG2N>
struct FftTwiddles {
    using O = PlanOps<LOG2N>;
    using P = typename O::P;
    float2 first[(1 << P::kR[0]) - 1];
    float2 last[O::kLastTwCount];
    const float2* mid;           // LDS

    // The middle pass's table goes to LDS through registers, a fixed number of loads per thread requested together
    // (index clamped: surplus threads repeat the last entry).  As a loop of load, wait, LDS store every workgroup began
    // with two or more exposed memory latencies.  request() asks for everything, park() -- any time before the first
    // exchange barrier, e.g. behind the request for the samples -- writes the LDS table.
    static constexpr int MID_PER = O::kMidCount > 0 ? (O::kMidCount + O::kWG - 1) / O::kWG : 0;
    float2 midv[MID_PER > 0 ? MID_PER : 1];

    __device__ __forceinline__ void park(int tid)
    {
        static_for<MID_PER>([&](auto I) {
            constexpr int k = I;
            int i = tid + k * O::kWG;
            if (i > O::kMidCount - 1) i = O::kMidCount - 1;
            const_cast<float2*>(mid)[i] = midv[k];
        });
    }

    // fills the LDS table (callers __syncthreads() before the first run) and the registers
    __device__ __forceinline__ void init(const float2* __restrict__ tw, float2* lds_mid, int tid, const FftAddr<LOG2N>& a)
    {
        request(tw, lds_mid, tid, a);
        park(tid);
    }

    __device__ __forceinline__ void request(const float2* __restrict__ tw, float2* lds_mid, int tid, const FftAddr<LOG2N>& a)
    {
        static_for<(1 << P::kR[0]) - 1>([&](auto I) { constexpr int i = I; first[i] = tw[i]; });
        static_for<MID_PER>([&](auto I) {
            constexpr int k = I;
            int i = tid + k * O::kWG;
            if (i > O::kMidCount - 1) i = O::kMidCount - 1;
            midv[k] = tw[O::kMidFirst + i];
        });
        mid = lds_mid;
        constexpr int LAST = O::kLast;
        const int jt = a.pos_last & ((1 << P::kC0[LAST]) - 1);
        static_for<P::kR[LAST]>([&](auto S) {
            constexpr int s = S;
            constexpr int c = P::kC0[LAST] + s;
            constexpr int mask = O::last_low_mask(c);
            static_for<(1 << O::popcount(mask))>([&](auto K) {
                constexpr int key = K;
                constexpr int jm = O::reg_pos(LAST, O::pdep(key, mask)) & ((1 << c) - 1);
                last[O::last_tw_offset(s) + key] = tw[(1 << c) - 1 + jm + jt];
            });
        });
    }
};

// All stages of the network on one symbol: x enters in the first-pass register pattern and
// leaves in the last-pass pattern (natural order bins).  Contains the workgroup barriers of the
// exchanges; the caller barriers before the exchange region is reused.
struct NoLateWork { __device__ __forceinline__ void operator()() const {} };

// `late` runs between the two barriers of the second exchange (3-pass plans): the place where
// kernels that are short of registers request what only the last pass and the epilogue need
// (last-pass twiddles, equaliser taps), so that it is not live across the first two passes.
// Data exchanged through LDS inside ONE wave needs program order only (LDS serves a wave's instructions in
// order), not a workgroup barrier: the fences keep the compiler from moving the reads above the stores
// (the bare wave barrier intrinsic carries no memory semantics).
__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int LOG2N, bool CONJ = false, typename Late = NoLateWork, int LATE_AT = 1>
__device__ __forceinline__ void fft_run(float2 (&x)[PlanOps<LOG2N>::kE], unsigned char* lds_data, const FftAddr<LOG2N>& a, const FftTwiddles<LOG2N>& w, Late&& late = Late{})
{
    using O = PlanOps<LOG2N>;
    using P = typename O::P;
    constexpr int E = O::kE, LAST = O::kLast;
    static_for<P::kR[0]>([&](auto S) {
        constexpr int c = P::kC0[0] + S;
        constexpr int bi = O::bfly_bit(0, c);
        static_for<E / 2>([&](auto H) {
            constexpr int m0 = O::insert_zero(H, bi), m1 = m0 | (1 << bi);
            constexpr int j = O::reg_pos(0, m0) & ((1 << c) - 1);
            bfly<CONJ, j == 0>(x[m0], x[m1], w.first[(1 << c) - 1 + j]);
        });
    });
    static_for<E>([&](auto M) { constexpr int m = M; *reinterpret_cast<float2*>(lds_data + a.wr0 + O::reg_addr(0, 0, m)) = x[m]; });
    if constexpr (LATE_AT == 0) late();
    __syncthreads();
    static_for<E>([&](auto M) { constexpr int m = M; x[m] = *reinterpret_cast<const float2*>(lds_data + a.rd0 + O::reg_addr(0, 1, m)); });
    if constexpr (O::kPasses == 3) {
        static_for<P::kR[1]>([&](auto S) {
            constexpr int c = P::kC0[1] + S;
            constexpr int bi = O::bfly_bit(1, c);
            static_for<E / 2>([&](auto H) {
                constexpr int m0 = O::insert_zero(H, bi), m1 = m0 | (1 << bi);
                constexpr int jm = O::reg_pos(1, m0) & ((1 << c) - 1);
                bfly<CONJ>(x[m0], x[m1], w.mid[(1 << c) - 1 - O::kMidFirst + jm + a.mid_jt]);
            });
        });
        __syncthreads();   // all reads of exchange 0 done before the region is rewritten
        if constexpr (LATE_AT == 1) late();
        static_for<E>([&](auto M) { constexpr int m = M; *reinterpret_cast<float2*>(lds_data + a.wr1 + O::reg_addr(1, 1, m)) = x[m]; });
        __syncthreads();
        static_for<E>([&](auto M) { constexpr int m = M; x[m] = *reinterpret_cast<const float2*>(lds_data + a.rd1 + O::reg_addr(1, 2, m)); });
    }
    if constexpr (LATE_AT == 2) late();      // behind the last exchange read: nothing but the last butterflies left to cover it
    static_for<P::kR[LAST]>([&](auto S) {
        constexpr int s = S;
        constexpr int c = P::kC0[LAST] + s;
        constexpr int bi = O::bfly_bit(LAST, c);
        constexpr int mask = O::last_low_mask(c);
        static_for<E / 2>([&](auto H) {
            constexpr int m0 = O::insert_zero(H, bi), m1 = m0 | (1 << bi);
            bfly<CONJ>(x[m0], x[m1], w.last[O::last_tw_offset(s) + O::pext(m0, mask)]);
        });
    });
    if constexpr (LATE_AT == 3) late();
}

typedef float sdr_v4f __attribute__((ext_vector_type(4)));
typedef float sdr_v2f __attribute__((ext_vector_type(2)));

// streaming (non-temporal) stores for outputs that are written once and never re-read here;
// only used where one wave instruction covers a contiguous span (see membench: scattered NT is slow)
__device__ __forceinline__ void stream_store4(float* dst, float a, float b, float c, float d)
{
    sdr_v4f v = {a, b, c, d};
    __builtin_nontemporal_store(v, reinterpret_cast<sdr_v4f*>(dst));
}
__device__ __forceinline__ void stream_store2(float* dst, float a, float b)
{
    sdr_v2f v = {a, b};
    __builtin_nontemporal_store(v, reinterpret_cast<sdr_v2f*>(dst));
}


// LLR staging through LDS: per-lane chunk of C floats -> lane-contiguous vectors (coalesced stores)
constexpr int stage_stride_bytes(int c) { return c == 6 ? 24 : c == 8 ? 48 : c == 12 ? 48 : c == 16 ? 80 : c == 20 ? 80 : c * 4; }
constexpr bool stage_wanted(int c) { return c == 6 || c == 8 || c == 10 || c == 12 || c == 16 || c == 20; }

__device__ __forceinline__ unsigned ref_bits(const uint8_t* __restrict__ s, size_t pos, int nbits)
{
    const size_t first = pos >> 3, last = (pos + nbits - 1) >> 3;
    unsigned acc = 0;
    for (size_t b = first; b <= last; ++b) acc = (acc << 8) | s[b];
    const unsigned span = static_cast<unsigned>(last - first + 1) * 8u;
    return (acc >> (span - static_cast<unsigned>(pos & 7) - static_cast<unsigned>(nbits))) & ((1u << nbits) - 1u);
}

__device__ __forceinline__ unsigned long long wave_sum_u64(unsigned long long v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// Compile-time shape of one instantiation: LDS layout, control structure and the bounds on the
// waves per SIMD handed to the compiler.
// NOWGT (EQ, LLR only): taps without a weight array -- the weight is the scalar w0 or is derived from the tap
// (RxParams::w_from_taps); no weight load is issued at all (the other EQ instantiations read a dummy source
// when an array is missing, to keep their loads unconditional).
template <int LOG2N, int MI, int MQ, bool FULL, bool EQ, bool HARDONLY, bool NOWGT = false, bool LC = false>
struct RxShape {
    using O = PlanOps<LOG2N>;
    static constexpr int N = O::kN, SPW = O::kSPW;
    static constexpr int BPS = MI + MQ;
    static constexpr bool DEMAP = BPS > 0;
    static_assert(!(HARDONLY && (FULL || EQ || !DEMAP)), "HARDONLY is a variant of its own");
    static_assert(!NOWGT || (EQ && !FULL && DEMAP), "taps without a weight array: a form of the LLR-only tap-reading variant");
    static_assert(!LC || (FULL && DEMAP && MI + MQ > 1), "LLRs + counters with nothing else: a form of the every-output variant");
    static constexpr int WG_THREADS = O::kWG;
    static constexpr int SYM_BYTES = O::max_sym_bytes();
    static constexpr int DATA_BYTES = SYM_BYTES * SPW;
    static constexpr int TW_BYTES = O::kMidCount * 8;
    using idx_t = std::conditional_t<(BPS > 8), uint16_t, uint8_t>;      // one hard decision
    // LC (round 3): the every-output variant with its outputs fixed at compile time -- LLRs written, hard decisions counted
    // against reference bits, no packed bits, no bins (the chunked link's receive step, include/sdr_amd.h
    // sdr_link_chain_c32): none of the per-group run-time tests of the output pointers, no decision staging in LDS.
    static constexpr bool HARD_PATH = (FULL || HARDONLY) && DEMAP && !LC;       // hard decisions pass through LDS and are packed
    // Hard-only kernels park their decisions in the exchange region, dead after the transform's last
    // exchange read (multi-wave workgroups pay one barrier for it): 10.2 -> 8.2 KiB at N = 1024.
    // The every-output variants do the same from N = 4096 on and at 1024-QAM (behind the staging strips,
    // which alias the region too): at N = 8192 it is the difference between one and two workgroups per CU (+12.5 to
    // +19.7 points), at N = 4096 +1.9 to +3.2, at 1024-QAM +1.5 to +3.3; the other shapes are level or lose the
    // price of the extra barrier (-0.4 to -1.9; same-buffer A/Bs, tools/abbench.py).
    static constexpr bool IDX_ALIAS = HARD_PATH && (HARDONLY || (FULL && (LOG2N >= 12 || BPS == 10)));
    static constexpr int IDX_RAW_BYTES = HARD_PATH ? N * SPW * static_cast<int>(sizeof(idx_t)) : 0;
    static constexpr int IDX_BYTES = IDX_ALIAS ? 0 : IDX_RAW_BYTES;
    // ... and park the reference bits there as well, written behind the last exchange read instead of ahead of the transform
    // (one more barrier): N = 8192 with 64-QAM and up would miss the second workgroup by 0.6 to 4.6 KiB otherwise.
    static constexpr bool REF_ALIAS = FULL && DEMAP && IDX_ALIAS;
    static constexpr int REF_SYM_BYTES = N * (DEMAP ? BPS : 8) / 8;      // reference bits of one symbol
    static constexpr int REF_RAW_BYTES = (FULL && DEMAP) ? REF_SYM_BYTES * SPW : 0;
    static constexpr int REF_BYTES = REF_ALIAS ? 0 : REF_RAW_BYTES;
    static constexpr int CHUNK = O::kGroupOut * BPS;                     // LLR floats per lane per output block
    // Several symbols per wave (N <= 256): one output block is a run of TPS * CHUNK floats per symbol.  Where that
    // run is not a whole number of 128-byte lines (N = 64 at QPSK / 64-QAM / 1024-QAM: 64 / 192 / 320 bytes) two
    // (BPSK: four) consecutive blocks are staged together, so that every store covers whole lines (128 / 384 / 640 bytes per symbol).
    // 16-QAM at N = 64 (one line per block, stored per lane) and QPSK / 1024-QAM at N = 256 take 256-byte runs:
    // +3.9 / +1.8 / +2.2 (with taps +2.3); 256-byte runs are level or worse for the other shapes.
    static constexpr int RUN_BYTES = O::kTPS * CHUNK * 4;
    static constexpr int stg_blocks()
    {
        if (!(DEMAP && !HARDONLY && SPW > 1 && O::kWG == 64 && stage_stride_bytes(CHUNK) == CHUNK * 4)) return 1;
        const int line = (CHUNK == 4 || (LOG2N == 8 && (BPS == 2 || BPS == 10))) ? 256 : 128;      // bytes a staged run must be a multiple of
        for (int nb = 1; nb <= 4; nb *= 2)
            if ((nb * RUN_BYTES) % line == 0 && (O::kE / O::kGroupOut) % nb == 0) return nb;
        return 1;
    }
    static constexpr int STG_BLOCKS = stg_blocks();
    static constexpr bool STAGE = DEMAP && !HARDONLY && (stage_wanted(CHUNK) || STG_BLOCKS > 1);
    static constexpr int STG_STRIDE = stage_stride_bytes(CHUNK);
    static constexpr int STG_WAVE_BYTES = STAGE ? 64 * STG_STRIDE * STG_BLOCKS : 0;
    static constexpr int STG_BYTES = STG_WAVE_BYTES * (O::kWG / 64);
    // The staging strips reuse the exchange region when they fit: it is dead between the last
    // exchange read and the next symbol's first exchange write.  Multi-wave workgroups need a
    // barrier for it, which costs more than it gains at N = 2048/4096 (measured) unless LDS is short:
    // the FULL variants there gain a workgroup per CU (4096/256-QAM + hard bits: 53 -> 59 %), and at
    // N = 8192 it is the difference between one and two workgroups per CU.
    static constexpr bool STG_ALIAS = STAGE && STG_BYTES <= SYM_BYTES * SPW && (O::kWG == 64 || LOG2N >= 13 || FULL);
    static constexpr int IDX_OFF = IDX_ALIAS ? (STG_ALIAS ? (STG_BYTES + 15) & ~15 : 0) : DATA_BYTES + TW_BYTES;
    static constexpr int REF_OFF = REF_ALIAS ? (IDX_OFF + IDX_RAW_BYTES + 15) & ~15 : DATA_BYTES + TW_BYTES + IDX_BYTES;
    static_assert(!IDX_ALIAS || IDX_OFF + IDX_RAW_BYTES <= DATA_BYTES, "aliased hard decisions fit the exchange region");
    static_assert(IDX_OFF % 16 == 0, "the packing step reads the decisions sixteen at a time");
    static_assert(!REF_ALIAS || REF_OFF + REF_RAW_BYTES <= DATA_BYTES, "aliased reference bits fit the exchange region");
    static constexpr int STG_OFF = STG_ALIAS ? 0 : (DATA_BYTES + TW_BYTES + IDX_BYTES + REF_BYTES + 15) & ~15;
    static constexpr int LDS_USED = (STG_ALIAS ? DATA_BYTES + TW_BYTES + IDX_BYTES + REF_BYTES : STG_OFF + STG_BYTES) + 16;
    // Resident workgroups per CU of the LLR-only kernels at N <= 1024.  These kernels are bound by the
    // memory system, and the memory system runs a 1 : bps/2 read/write mix FASTER with fewer symbols
    // in flight (a copy of this shape: 5.4 TB/s at 16 waves per CU, 6.0 at 4-5, tools/membench6/7),
    // while the kernel needs enough waves to cover its own latency: the optimum sits at 6-12
    // workgroups per CU instead of the 16-32 that LDS and registers would allow.  Measured on one set
    // of buffers per case (tools/occbench.py, profiles/r2_occupancy_single_allocation.txt): +2 to +5
    // points.  The throttle is unused LDS; 0 = no throttle (N >= 2048, QPSK at 1024 and the variants
    // with hard bits or counters measured best at what they already get).
    static constexpr int wg_per_cu_target()
    {
        if (FULL || EQ || HARDONLY || !DEMAP || BPS == 1) return 0;      // BPSK (2:1 read-dominated) was not part of the sweep
        switch (LOG2N) {
            case 6: return BPS <= 8 ? (BPS == 4 ? 9 : 8) : 0;
            case 7: return BPS <= 4 ? 6 : BPS == 6 ? 7 : BPS == 8 ? 8 : 0;
            case 8: return BPS == 2 ? 0 : BPS == 4 ? 6 : BPS == 6 ? 8 : BPS == 8 ? 10 : 0;      // QPSK: -2.1 with the throttle (same-buffer A/B)
            case 9: return BPS == 2 ? 14 : BPS <= 8 ? 12 : 0;
            case 10: return BPS == 4 ? 10 : BPS == 6 ? 9 : BPS == 8 ? 12 : 0;
            default: return 0;
        }
    }
    static constexpr int LDS_THROTTLE = wg_per_cu_target() > 0 ? ((160 * 1024 / wg_per_cu_target()) & ~511) : 0;
    static constexpr int LDS_BYTES = LDS_USED > LDS_THROTTLE ? LDS_USED : LDS_THROTTLE;

    // How the LLR-only tap-holding variant gets its taps, per shape, from same-buffer A/Bs of two builds inside
    // one process (tools/abbench.py, profiles/r2_eq_straight_line_ab.txt):
    //   0  run of two symbols (seven where RUN says so), taps held in registers and requested ahead of the samples when the set changes
    //   1  straight-line, one symbol group per workgroup, taps requested unconditionally behind the samples
    //   2  straight-line, taps requested per symbol ahead of the first exchange barrier (LATE_TAPS)
    //   3  straight-line, taps read in the epilogue EPI_AHEAD register groups ahead of their use (EPILOGUE_TAPS; N = 8192, 1024/1024-QAM)
    static constexpr int eq_form()
    {
        if (BPS == 1) return LOG2N == 13 ? 3 : 0;
        switch (LOG2N) {
            case 6: return BPS == 2 ? 0 : 1;
            case 7: return BPS == 10 ? 0 : 1;
            case 8: return (BPS <= 4 || BPS == 10) ? 0 : 1;
            case 10: return BPS == 2 ? 0 : BPS == 10 ? 3 : 1;      // 1024-QAM: +3.2 with the epilogue form and its prefetch (the form loses 0.2 to 8.5 in every other cell below N = 8192)
            case 11: return (BPS == 2 || BPS == 10) ? 0 : BPS == 4 ? 1 : 2;
            case 12: return (BPS <= 4 || BPS == 10) ? 0 : 2;       // 1024-QAM: the run of two again since the unpacked build (-2.2 with the straight form)
            case 13: return 3;
            default: return 0;      // N = 512: the run of two
        }
    }
    // Form 2: requested per symbol ahead of the first exchange barrier -- 48 fewer live registers, three
    // waves per SIMD instead of two, but twice the tap traffic through L2.  It wins at N = 2048 (64-QAM up) and 4096
    // (profiles/r2_eq_straight_line_ab.txt).  Four waves per SIMD spill (-15 points).
    static constexpr bool LATE_TAPS = EQ && DEMAP && O::kPasses == 3 && !FULL && eq_form() == 2;
    static constexpr int LATE_AT = 0;
    // Only the tap-holding instantiations take a run of symbol groups per workgroup (taps reloaded
    // once per run).  The others take exactly one group and are straight-line code: nothing is
    // hoisted out of a loop into registers (N = 1024 LLR variant: 136 -> 104-119 registers; hard bits
    // beside LLRs: 63 -> 68 %; N = 8192: 56 -> 71 %).  Counting variants then end every symbol with
    // four atomics per wave on the 1024 counter slots, which measured no slower than runs of eight.
    // Form 1: the tap-holding variant is straight-line too -- one symbol group per workgroup, taps and weights
    // requested unconditionally BEHIND the samples, so that the wait in front of the transform covers the
    // samples only (vmcnt counts the tap loads as still outstanding) and the taps land during it.  Twice
    // the tap requests through L2, but three waves per SIMD instead of two and no loads queued behind a
    // previous symbol's stores: +1 to +7 points at N = 64...256, 1024 (16-QAM up), 2048, 4096; N = 512 and
    // QPSK at 1024 lose 0.4 to 6.7 points with it and N = 8192 is level: those keep the run of two.
    // Every-output variants (hard bits, counters beside LLRs; same tool, N:bps:eq:hard): straight wins +1.5 to +7.1 at
    // 64 / 256 from 64-QAM, 256 everywhere, 1024 at 256-QAM, 2048 from 16-QAM to 256-QAM; loses 1.4 to 5.4 at 512, 4096, QPSK
    static constexpr bool full_straight()
    {
        switch (LOG2N) {
            case 6: case 7: return BPS >= 6 && BPS <= 8;
            case 8: return BPS >= 4 && BPS <= 8;
            case 10: return BPS == 8;
            case 11: return BPS >= 6 && BPS <= 8;
            default: return false;
        }
    }
    static constexpr bool EQ_STRAIGHT_RULE = FULL ? full_straight() : (eq_form() == 1);
    // Form 3.  N = 8192 (E = 32): 96 tap registers beside 64 of data and the last pass's twiddles do not fit 256
    // (one wave per SIMD, or spills with late taps); there the epilogue reads each register group's taps
    // and weights EPI_AHEAD groups ahead of where it uses them (+0.2 to +9.5 points over one wave per SIMD with held taps).
    // The every-output variants with taps take the same form from N = 4096 on, at N = 1024 and at 1024-QAM:
    // held taps cost them 50 to 100 spilled registers at N = 8192 with two waves per SIMD (+3.4 to +14.8 points,
    // same-buffer A/B); N = 4096: +2.5 to +6.7; 1024-QAM: -0.1 to +5.2; N = 1024: +0.6 to +4.9; the other shapes: -1.7 to +1.1, left alone.
    static constexpr bool EPILOGUE_TAPS = EQ && DEMAP && (FULL ? (LOG2N >= 12 || BPS == 10 || LOG2N == 10) : eq_form() == 3);
    // taps of register group h + EPI_AHEAD requested before group h is worked on: same-buffer A/B at N = 8192: BPSK 32.4 -> 49.3 %,
    // QPSK 42.7 -> 56.6, 16-QAM 50.3 -> 60.0, the others +0.2 to +1.0 (8: level, 1024-QAM -1.6)
    static constexpr int EPI_AHEAD = 4;
    // Non-temporal sample loads in the tap-reading variants: the samples are read once, the taps of a set
    // by up to 14 workgroups.  Same-buffer A/B (tools/abbench.py): +0.4 to +2.5 points at N = 1024, +0.9 to
    // +11.3 at N = 8192 (64-QAM: 47.7 -> 59.0 %), level (-0.5 ... +0.2) at the other sizes, which keep plain
    // loads; the variants without taps lose up to 1.7 points with them.
    static constexpr bool NT_SAMPLES = EQ && DEMAP && (LOG2N == 10 || LOG2N == 13);
    static constexpr bool EQ_STRAIGHT = EQ && DEMAP && !LATE_TAPS && EQ_STRAIGHT_RULE;
    static constexpr bool TAPS_AFTER = EQ_STRAIGHT;
    // Symbol groups per workgroup of the loop form (taps reloaded when the set changes, i.e. once per run inside a set).
    // Seven, as a real loop, where the taps' way through L2 weighs most against the symbol's own bytes (same-buffer
    // A/B against the run of two: N = 4096 BPSK +4.9, QPSK +4.2, 1024-QAM +1.3; N = 2048 +2.3 / +1.6 / +0.8);
    // it loses 0.4 to 4.5 points at N <= 1024 and at 4096/16-QAM, and fourteen is no better than seven.
    // QPSK at N = 64 / 128 without taps: the smallest symbols of all (512 / 1024 bytes of LLRs), where a workgroup's set-up weighs most --
    // a run of two (N = 64) or three (N = 128) symbol groups per workgroup: 63.1 -> 72.5 % and 69.0 -> 70.9 % (same-buffer A/B,
    // profiles/r3_small_n_qpsk_runs_ab.txt); every other small shape (BPSK, 16-QAM up, N = 256) loses 1 to 11 points with runs.
    static constexpr bool SMALL_LOOP = (LOG2N == 6 || LOG2N == 7) && BPS == 2 && !EQ && !FULL && !HARDONLY;
    static constexpr int RUN = SMALL_LOOP ? (LOG2N == 6 ? 2 : 3) : ((LOG2N == 11 || LOG2N == 12) && (BPS <= 2 || BPS == 10)) ? 7 : 2;
    static constexpr bool LOOP = (EQ && !LATE_TAPS && !EQ_STRAIGHT && !EPILOGUE_TAPS) || SMALL_LOOP;
    // The middle pass's twiddle table is written to LDS BEHIND the request for the first symbol's samples: the
    // workgroup then waits for one memory round trip instead of two in a row (table, then samples).  The kernels
    // bound by HBM lose 1.8 to 6.1 points with it at N = 1024 ... 4096 (same-buffer A/B, round 2) and keep the set-up's
    // own wait; the hard-only kernel, whose waves spent 35 % of their life in s_waitcnt with two such round
    // trips per symbol (profiles/r3_pmc_hard_only_65536sym.txt), takes it: +0.9 to +1.2 points at N = 512 ... 8192
    // (profiles/r3_hard_only_table_late_and_prefetch_ab.txt; the counting variants +0.2 to +0.5, left alone).
    static constexpr bool TW_LATE = HARDONLY;
    // The XCD-aware workgroup order makes each XCD's STORES one sequential stream.  The hard-only kernel reads ten
    // times what it writes and runs 0.5 to 1.5 points faster in the natural order (same file).
    static constexpr bool XCD_ORDER = !HARDONLY;

    // Waves per SIMD.  Lower bound = register budget: the tap-holding and FULL variants must not
    // drift to 256+ registers (one wave per SIMD); at N = 8192 (E = 32) FULL variants are held to
    // one workgroup per CU by LDS and tap-holding ones would spill at two (measured 49 -> 33 %).
    // Upper bound = what LDS lets a CU hold, spread evenly over its four SIMDs: when the registers
    // allow more per SIMD than that, the dispatcher fills the SIMDs unevenly and a two- or four-wave
    // workgroup loses 5-8 points (N = 2048 / 64-QAM at 114 registers: 65 %; the same code padded to
    // 168 registers, i.e. three waves per SIMD: 73 %).  The compiler pads the allocation for it.
    static constexpr int WAVES_PER_WG = WG_THREADS / 64;
    static constexpr int LDS_WAVES = (160 * 1024 / ((LDS_BYTES + 511) & ~511)) * WAVES_PER_WG;
    static constexpr int MIN_WAVES = (HARDONLY && LOG2N <= 10) ? 1
                                     : LOG2N >= 13 ? (EPILOGUE_TAPS ? 2 : (FULL && IDX_ALIAS) ? 2 : (FULL || EQ || HARDONLY) ? 1 : 2)
                                     : FULL ? 2 : EQ ? (LATE_TAPS ? 3 : 2) : 1;
    static constexpr int even_share() { const int w = (LDS_WAVES + 3) / 4; return w < 1 ? 1 : w > 8 ? 8 : w; }
    static constexpr int MAX_WAVES = even_share() < MIN_WAVES ? MIN_WAVES : even_share();
};

// MI + MQ == 0: FFT only (bins to p.fft).  FULL: every output optional at run time.
// EQ: per-subcarrier taps / LLR weights may be given (held in registers across the symbols of a
// set, or requested per symbol: RxShape::LATE_TAPS; costs registers, hence a separate instantiation).
// HARDONLY: packed hard decisions are the only output (the reference's own chain ofdm::rx ->
// from_constl, lib/inc/ofdm.hpp:37-43 + modulation.hpp:96-112): no LLR arithmetic at all.
template <int LOG2N, int MI, int MQ, bool FULL, bool EQ, bool HARDONLY = false, bool NOWGT = false, bool LC = false>
__global__ void __launch_bounds__(PlanOps<LOG2N>::kWG)
    __attribute__((amdgpu_waves_per_eu(RxShape<LOG2N, MI, MQ, FULL, EQ, HARDONLY, NOWGT, LC>::MIN_WAVES, RxShape<LOG2N, MI, MQ, FULL, EQ, HARDONLY, NOWGT, LC>::MAX_WAVES)))
    rx_fused_kernel(const RxParams p)
{
    using O = PlanOps<LOG2N>;
    using P = typename O::P;
    using S = RxShape<LOG2N, MI, MQ, FULL, EQ, HARDONLY, NOWGT, LC>;
    using idx_t = typename S::idx_t;
    constexpr int N = O::kN, E = O::kE, TPS = O::kTPS, SPW = O::kSPW, LAST = O::kLast;
    constexpr int BPS = S::BPS;
    constexpr bool DEMAP = S::DEMAP, HARD_PATH = S::HARD_PATH, STAGE = S::STAGE, STG_ALIAS = S::STG_ALIAS;
    constexpr int SYM_BYTES = S::SYM_BYTES, DATA_BYTES = S::DATA_BYTES;
    constexpr int REF_SYM_BYTES = S::REF_SYM_BYTES, CHUNK = S::CHUNK, STG_STRIDE = S::STG_STRIDE;
    constexpr int STG_WAVE_BYTES = S::STG_WAVE_BYTES, STG_OFF = S::STG_OFF, LDS_BYTES = S::LDS_BYTES;
    constexpr bool LATE_TAPS = S::LATE_TAPS, LOOP = S::LOOP;
    // The LLR-only tap-reading kernels are only dispatched with the arrays they read (taps and weights, or taps alone for
    // NOWGT; a call with weights alone takes the every-output instantiation): no run-time test of p.eq / p.wgt, which the
    // compiler turned into three selects per bin with both operands live.
    constexpr bool TAPS_SURE = EQ && !FULL, WGT_SURE = EQ && !FULL && !NOWGT;
    // which outputs this call produces: run-time pointers, or fixed by the LC instantiation
    auto want_llr = [&]() { if constexpr (LC) return true; else return p.llr != nullptr; };
    auto want_fft = [&]() { if constexpr (LC) return false; else return p.fft != nullptr; };
    auto want_counts = [&]() { if constexpr (LC) return true; else return p.counters != nullptr; };
    __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];
    unsigned char* lds_data = lds;
    [[maybe_unused]] idx_t* lds_idx = reinterpret_cast<idx_t*>(lds + S::IDX_OFF);
    [[maybe_unused]] unsigned char* lds_ref = lds + S::REF_OFF;

    const int tid = static_cast<int>(threadIdx.x);
    const int slot = SPW == 1 ? 0 : tid / TPS;                            // exchange region of the thread's symbol
    const int t = SPW == 1 ? tid : tid % TPS;
    constexpr int SPG = SPW;                                              // symbols per symbol group

    // ---- per-workgroup set-up: addresses and twiddles ----
    FftAddr<LOG2N> addr;
    addr.init(t, slot, SYM_BYTES);
    FftTwiddles<LOG2N> tw;
    // the middle pass's table: all of a thread's entries requested together, then written to LDS (one wait instead of a
    // load-wait-store loop: level, -0.6 ... +0.9); S::TW_LATE: written behind the request for the first symbol's samples
    tw.request(p.tw, reinterpret_cast<float2*>(lds + DATA_BYTES), tid, addr);
    [[maybe_unused]] bool tw_parked = !S::TW_LATE;
    if constexpr (!S::TW_LATE) tw.park(tid);
    const int pos_last = addr.pos_last;
    const int in_thread = addr.in_thread;

    unsigned long long c_bit = 0, c_sym = 0, c_abs = 0, c_sign = 0;

    const unsigned long long n_groups = (p.n_sym + SPG - 1) / SPG;
    const size_t in_stride = static_cast<size_t>(N) + p.cp;
    if constexpr (O::kMidCount > 0 && !S::TW_LATE) __syncthreads();

    // global load of one symbol into registers: CP skipped, bit reversal folded into the index
    auto load_symbol = [&](unsigned long long grp_, float2 (&dst)[E]) {
        unsigned long long s_ = grp_ * SPG + slot;
        if (s_ >= p.n_sym) s_ = p.n_sym - 1;
        const float2* src = p.rx + s_ * in_stride + p.cp + in_thread;
        constexpr int GI = O::kGroupIn;
        static_assert(GI == 1 || GI == 2, "plans load one or two adjacent samples per register group");
        static_for<E / GI>([&](auto Q) {
            constexpr int q = Q;   // butterfly register bits of pass 0 (low kR[0] register bits)
            constexpr int idx0 = O::bitrev(O::reg_pos(0, q), LOG2N);
            if constexpr (GI == 1) {
                dst[q] = src[idx0];
            } else {
                // register bit kR[0] is the group bit: adjacent input samples
                float4 v;
                if constexpr (S::NT_SAMPLES) {
                    const sdr_v4f q = __builtin_nontemporal_load(reinterpret_cast<const sdr_v4f*>(src + idx0));
                    v = make_float4(q.x, q.y, q.z, q.w);
                } else {
                    v = *reinterpret_cast<const float4*>(src + idx0);
                }
                dst[q] = make_float2(v.x, v.y);
                dst[q | (1 << P::kR[0])] = make_float2(v.z, v.w);
            }
        });
    };

    // Workgroups are dealt round-robin over the 8 XCDs (blockIdx % 8 shares an XCD).  Inside
    // windows of xcd_window workgroups the ids are permuted so that each XCD works on a run of
    // CONSECUTIVE symbols: its stores then form one sequential stream instead of every eighth
    // chunk (measured +6..10 % on the write path, tools/membench2/3).  The default window is the
    // whole grid (one contiguous eighth of the batch per XCD: +1 to +3 points over windows of 4096
    // on 1 Mi-symbol batches, tools/sweep_xcd_window.sh).  Speed only, never correctness.
    unsigned long long wg = blockIdx.x;
    if (S::XCD_ORDER && p.xcd_window >= 8) {
        const unsigned long long win = p.xcd_window;
        const unsigned long long full = (static_cast<unsigned long long>(gridDim.x) / win) * win;
        if (wg < full) {
            const unsigned long long r = wg % win;
            const unsigned long long share = win >> 3;
            unsigned long long j = (r >> 3) + (r & 7) * p.xcd_stagger;      // launch_rx_fused keeps 8 * stagger below the share
            if (j >= share) j -= share;
            wg = wg - r + (r & 7) * share + j;
        }
    }
    const unsigned long long grp_begin = wg * p.groups_per_wg;
    const unsigned long long grp_end = (grp_begin + p.groups_per_wg < n_groups) ? grp_begin + p.groups_per_wg : n_groups;
    float2 x[E];
    constexpr bool PREFETCH_TAPS = EQ && DEMAP && !S::EPILOGUE_TAPS;
    [[maybe_unused]] float2 eqv[PREFETCH_TAPS ? E : 1];
    [[maybe_unused]] float wv[(PREFETCH_TAPS && !NOWGT) ? E : 1];
    [[maybe_unused]] size_t held_set = ~static_cast<size_t>(0);     // set whose taps the registers hold
    if (grp_begin >= grp_end) return;      // uniform over the workgroup
    // One symbol group, as a body that the tap-holding variants run twice: written out, because a
    // loop around barriers does not unroll and anything hoisted out of a rolled loop costs
    // registers the transform needs (243 -> 180 at N = 2048).
    auto process = [&](const unsigned long long grp) __attribute__((always_inline)) {
        unsigned long long sym = grp * SPG + slot;
        const bool active = sym < p.n_sym;
        if (!active) sym = p.n_sym - 1;

        // Taps and weights of this thread's output bins, kept in registers for as long as the
        // run stays inside one set (a workgroup's symbols are consecutive, a set spans
        // sym_per_set of them).  When the set changes they are requested FIRST, ahead of the
        // samples: vector-memory results return in order, so the wait for the samples (which the
        // transform needs anyway) covers them and nothing is left to wait for in the epilogue.
        // A load under a run-time branch placed AFTER the samples would make the compiler's
        // wait-count bookkeeping fall back to "wait for everything" in front of the transform.
        // A missing array is replaced by an always-valid dummy source (this symbol's own
        // samples) whose values are never used.
        [[maybe_unused]] size_t tap_set = 0;
        if constexpr (EQ && DEMAP) tap_set = p.sym_per_set ? static_cast<size_t>(sym / p.sym_per_set) : 0;
        auto load_taps = [&]() {
            if constexpr (PREFETCH_TAPS) {
            constexpr int GO_ = O::kGroupOut;
            const float2* dummy = p.rx + static_cast<size_t>(sym) * in_stride;      // >= N readable float2
            const float2* eq_src = (TAPS_SURE || p.eq) ? p.eq + tap_set * N : dummy;
            [[maybe_unused]] const float* wgt_src = (WGT_SURE || p.wgt) ? p.wgt + tap_set * N : reinterpret_cast<const float*>(dummy);
            static_for<E / GO_>([&](auto Hh) {
                constexpr int h = Hh;
                const int k = O::reg_pos(LAST, h * GO_) | pos_last;
                if constexpr (GO_ == 2) {
                    const float4 v = *reinterpret_cast<const float4*>(eq_src + k);
                    eqv[h * 2] = make_float2(v.x, v.y);
                    eqv[h * 2 + 1] = make_float2(v.z, v.w);
                    if constexpr (!NOWGT) {
                        const float2 u = *reinterpret_cast<const float2*>(wgt_src + k);
                        wv[h * 2] = u.x;
                        wv[h * 2 + 1] = u.y;
                    }
                } else {
                    static_for<GO_>([&](auto G) { constexpr int g = G; eqv[h * GO_ + g] = eq_src[k + g]; if constexpr (!NOWGT) wv[h * GO_ + g] = wgt_src[k + g]; });
                }
            });
            }
        };
        if constexpr (PREFETCH_TAPS && !LATE_TAPS && !S::TAPS_AFTER) {
            if (tap_set != held_set) {
                held_set = tap_set;
                load_taps();
            }
        }

        // Reference bits of the symbol (counting only): requested ahead of the samples, parked
        // in LDS before the transform.  Unconditional (see the taps above); a
        // missing array is replaced by the symbol's own samples, never looked at.
        constexpr int REF_WORDS = REF_SYM_BYTES / 4;
        constexpr int REF_PER_THREAD = (REF_WORDS + TPS - 1) / TPS;
        [[maybe_unused]] uint32_t refw[REF_PER_THREAD];
        if constexpr (FULL && DEMAP) {
            const uint32_t* ref_src = (LC || p.ref) ? reinterpret_cast<const uint32_t*>(p.ref + static_cast<size_t>(sym) * REF_SYM_BYTES)
                                            : reinterpret_cast<const uint32_t*>(p.rx + static_cast<size_t>(sym) * in_stride);
            static_for<REF_PER_THREAD>([&](auto R) {
                constexpr int r = R;
                int i = r * TPS + t;
                if (i > REF_WORDS - 1) i = REF_WORDS - 1;
                refw[r] = ref_src[i];
            });
        }

        // ---- pass 0 ----
        load_symbol(grp, x);
        if constexpr (S::TW_LATE) if (!tw_parked) { tw.park(tid); tw_parked = true; }      // the exchange barriers of pass 0 lie between this and the table's first use
        if constexpr (PREFETCH_TAPS && !LATE_TAPS && S::TAPS_AFTER) load_taps();      // straight-line: unconditional, behind the samples
        auto park_ref = [&]() {
            static_for<REF_PER_THREAD>([&](auto R) {
                constexpr int r = R;
                int i = r * TPS + t;
                if (i > REF_WORDS - 1) i = REF_WORDS - 1;          // duplicates store the same word
                reinterpret_cast<uint32_t*>(lds_ref)[slot * REF_WORDS + i] = refw[r];
            });
        };
        if constexpr (FULL && DEMAP && !S::REF_ALIAS) park_ref();

        if constexpr (LATE_TAPS) {
            auto late = [&]() { load_taps(); };
            fft_run<LOG2N, false, decltype(late), S::LATE_AT>(x, lds_data, addr, tw, std::move(late));
        } else {
            fft_run<LOG2N>(x, lds_data, addr, tw);
        }
        // every wave is past its last exchange read (staging strips alias it)
        if constexpr ((STG_ALIAS || S::IDX_ALIAS) && O::kWG > 64) __syncthreads();
        if constexpr (S::REF_ALIAS) {      // reference bits into the dead exchange region, visible to the symbol's other threads
            park_ref();
            if constexpr (O::kWG > 64) __syncthreads(); else wave_lds_sync();
        }

        // ---- epilogue: bins are in natural order, k = reg_pos(LAST, m) | pos_last ----
        constexpr int GO = O::kGroupOut;   // adjacent bins held in registers m = GO*h + (0..GO-1)
        const size_t bin0 = static_cast<size_t>(sym) * N;
        // Bins out with several symbols per wave whose per-block run (TPS * GO bins) is half a cache line (N = 64): two
        // consecutive blocks go through the dead exchange region (one-wave workgroups: LDS ops are in order) so that
        // every store covers whole 128-byte lines.
        constexpr bool BINS_STAGED = SPW > 1 && O::kWG == 64 && (TPS * GO * 8) % 128 != 0 && (2 * TPS * GO * 8) % 128 == 0
                                     && (E / GO) % 2 == 0 && 64 * GO * 2 * 2 * 4 <= DATA_BYTES;
        if constexpr ((!DEMAP || FULL) && BINS_STAGED) {
            if (want_fft()) {
                constexpr int CB = GO * 2, NBB = 2;                      // floats per lane and block, blocks per group
                float* stg = reinterpret_cast<float*>(lds_data);
                const int lane = tid & 63;
                static_for<E / GO>([&](auto Hh) {
                    constexpr int h = Hh, hb = h % NBB, h0 = h - hb;
                    static_assert(O::reg_pos(LAST, (h0 + 1) * GO) == O::reg_pos(LAST, h0 * GO) + TPS * GO, "paired blocks are adjacent in memory");
                    float* mine = stg + (((lane / TPS) * NBB + hb) * TPS + (lane % TPS)) * CB;
                    static_for<GO>([&](auto G) { constexpr int g = G; *reinterpret_cast<float2*>(mine + 2 * g) = x[h * GO + g]; });
                    if constexpr (hb == NBB - 1) {
                        wave_lds_sync();      // one wave: LDS serves its instructions in order, the fences keep the compiler from reordering them
                        constexpr int RUN = TPS * CB * NBB;                  // floats per symbol in the group
                        constexpr int NV = 64 * CB * NBB / 4;
                        static_assert(RUN % 4 == 0 && NV % 64 == 0, "whole vectors, whole wave");
                        static_for<NV / 64>([&](auto I) {
                            constexpr int i = I;
                            const int v = i * 64 + lane;
                            const int s_ = (v * 4) / RUN, off = (v * 4) % RUN;
                            const unsigned long long dsym = grp * SPW + s_;
                            if (dsym < p.n_sym) {
                                const float4 q4 = *reinterpret_cast<const float4*>(stg + v * 4);
                                *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.fft + static_cast<size_t>(dsym) * N + O::reg_pos(LAST, h0 * GO)) + off) = q4;
                            }
                        });
                        wave_lds_sync();      // one wave: LDS serves its instructions in order, the fences keep the compiler from reordering them
                    }
                });
            }
        } else if constexpr (!DEMAP || FULL) {
            if (want_fft() && active) {
                static_for<E / GO>([&](auto Hh) {
                    constexpr int h = Hh;
                    const int k = O::reg_pos(LAST, h * GO) | pos_last;
                    if constexpr (GO == 2) {
                        *reinterpret_cast<float4*>(p.fft + bin0 + k) = make_float4(x[h * GO].x, x[h * GO].y, x[h * GO + 1].x, x[h * GO + 1].y);
                    } else {
                        static_for<GO>([&](auto G) { constexpr int g = G; p.fft[bin0 + k + g] = x[h * GO + g]; });
                    }
                });
            }
        }
        // hard decisions of the symbol group, parked in lds_idx, leave as MSB-first words
        [[maybe_unused]] auto pack_hard = [&]() {
            if constexpr (HARD_PATH) if (HARDONLY || p.hard) {
                __syncthreads();
                if (active) pack_decisions<N, BPS, idx_t>(lds_idx + slot * N, p.hard + bin0 * BPS / 8, t, TPS);
            }
        };
        // Hard decisions alone: the margin slicer (qam.hpp) runs over ALL of the thread's bins as one
        // branch-free block -- a per-point branch to the out-of-line test costs eight scalar instructions, two exec
        // round trips and a scheduling barrier per point -- and keeps, per register group, the mask of lanes whose
        // proposal was not clearly inside its cell (one v_cmp into a scalar pair).  Waves with any such lane (about one
        // in four on noise: a point in 4000 is doubtful) then redo exactly those groups for exactly those lanes with the
        // three-point test / scan and overwrite the parked decisions.  From 16-QAM up at every size; BPSK and QPSK (whose scan is
        // four points) from N = 128: QPSK +1 to +3.7 there, -1.3 at N = 64, BPSK level (same-buffer A/B).
        constexpr bool SLICE_BLOCK = HARDONLY && (MI >= 2 || MQ >= 2 || LOG2N >= 7) && !SDR_MARGIN_OFF;
        if constexpr (SLICE_BLOCK) {
            using MS = MarginSlicer<float, MI, MQ>;
            constexpr int NG = E / GO;
            const float half = p.inorm * 0.5f;
            const float limit = MS::reach_limit(half);
            bool doubt[NG];      // per lane; the compiler keeps each as a scalar pair (the compares' own results)
            bool any = false;
            static_for<NG>([&](auto Hh) {
                constexpr int h = Hh;
                const int k = O::reg_pos(LAST, h * GO) | pos_last;
                bool ok = true;
                float w[GO];      // level_I * 2^MQ + level_Q, an exact small integer
                static_for<GO>([&](auto G) {
                    constexpr int g = G;
                    float miss, reach;
                    w[g] = MS::propose(x[h * GO + g].x, x[h * GO + g].y, half, miss, reach);
                    ok = static_cast<bool>(static_cast<int>(ok) & static_cast<int>(MS::sure(miss, reach, limit)));
                });
                if constexpr (GO == 2 && sizeof(idx_t) == 1) {
                    // both points combined (exactly, in float) and Gray-coded in one go: the bytes keep their order in LDS
                    // (little endian: bin k first)
                    constexpr unsigned keep1 = 0x7fu & (MQ > 0 ? ~(1u << (MQ - 1)) : ~0u);
                    const unsigned pair = static_cast<unsigned>(static_cast<int>(__builtin_fmaf(w[1], 256.0f, w[0])));
                    const unsigned coded = pair ^ ((pair >> 1) & (keep1 | (keep1 << 8)));
                    *reinterpret_cast<uint16_t*>(lds_idx + slot * N + k) = static_cast<uint16_t>(coded);
                } else {
                    static_for<GO>([&](auto G) { constexpr int g = G; lds_idx[slot * N + k + g] = static_cast<idx_t>(MS::gray(static_cast<unsigned>(static_cast<int>(w[g])))); });
                }
                doubt[h] = !ok;
                any = static_cast<bool>(static_cast<int>(any) | static_cast<int>(doubt[h]));
            });
            if (__builtin_amdgcn_ballot_w64(any) != 0) {                // uniform over the wave
                static_for<NG>([&](auto Hh) {
                    constexpr int h = Hh;
                    if (doubt[h]) {
                        const int k = O::reg_pos(LAST, h * GO) | pos_last;
                        static_for<GO>([&](auto G) {
                            constexpr int g = G;
                            lds_idx[slot * N + k + g] = static_cast<idx_t>(nearest_checked<float, MI, MQ>(x[h * GO + g].x * p.inorm, x[h * GO + g].y * p.inorm));
                        });
                    }
                });
            }
            pack_hard();
        } else
        if constexpr (DEMAP) {
            // EPILOGUE_TAPS: the taps of register group h + AHEAD are requested before group h
            // is worked on (the run-time branches on p.eq / p.wgt and the staging barriers end the compiler's scheduling
            // regions, so it hoists nothing across groups by itself: one exposed memory latency per group).
            constexpr int NBLK = E / GO;
            constexpr int AHEAD = S::EPILOGUE_TAPS ? (S::EPI_AHEAD < NBLK ? S::EPI_AHEAD : NBLK) : 0;
            [[maybe_unused]] float2 epv[AHEAD > 0 ? E : 1];
            [[maybe_unused]] float epw[(AHEAD > 0 && !NOWGT) ? E : 1];
            [[maybe_unused]] auto epi_load = [&]<int H_>() {
                const float2* dummy = p.rx + static_cast<size_t>(sym) * in_stride;
                const float2* eq_src = (TAPS_SURE || p.eq) ? p.eq + tap_set * N : dummy;
                [[maybe_unused]] const float* wgt_src = (WGT_SURE || p.wgt) ? p.wgt + tap_set * N : reinterpret_cast<const float*>(dummy);
                const int k = O::reg_pos(LAST, H_ * GO) | pos_last;
                if constexpr (GO == 2) {
                    const float4 v = *reinterpret_cast<const float4*>(eq_src + k);
                    epv[H_ * 2] = make_float2(v.x, v.y); epv[H_ * 2 + 1] = make_float2(v.z, v.w);
                    if constexpr (!NOWGT) {
                        const float2 u = *reinterpret_cast<const float2*>(wgt_src + k);
                        epw[H_ * 2] = u.x; epw[H_ * 2 + 1] = u.y;
                    }
                } else {
                    static_for<GO>([&](auto G) { constexpr int g = G; epv[H_ * GO + g] = eq_src[k + g]; if constexpr (!NOWGT) epw[H_ * GO + g] = wgt_src[k + g]; });
                }
            };
            if constexpr (AHEAD > 0) static_for<AHEAD>([&](auto Hh) { constexpr int h = Hh; epi_load.template operator()<h>(); });
            static_for<E / GO>([&](auto Hh) {
                constexpr int h = Hh;
                const int k = O::reg_pos(LAST, h * GO) | pos_last;
                float l[GO * BPS];
                unsigned idx[GO];
                if constexpr (AHEAD > 0 && h + AHEAD < NBLK) epi_load.template operator()<h + AHEAD>();
                // EPILOGUE_TAPS: this register group's taps and weights, read unconditionally (dummy source when an
                // array is missing) so that the compiler can count and hoist the loads
                [[maybe_unused]] float2 epi_eq[GO];
                [[maybe_unused]] float epi_w[GO];
                if constexpr (S::EPILOGUE_TAPS) {
                    static_for<GO>([&](auto G) { constexpr int g = G; epi_eq[g] = epv[h * GO + g]; if constexpr (!NOWGT) epi_w[g] = epw[h * GO + g]; });
                }
                static_for<GO>([&](auto G) {
                    constexpr int g = G;
                    float2 y = x[h * GO + g];
                    float w = p.w0;
                    if constexpr (EQ) {
                        if (TAPS_SURE || p.eq) {
                            float2 gk;
                            if constexpr (PREFETCH_TAPS) gk = eqv[h * GO + g];
                            else gk = epi_eq[g];
                            const float er = y.x * gk.x - y.y * gk.y;
                            const float ei = y.x * gk.y + y.y * gk.x;
                            y.x = er; y.y = ei;
                            // no weight array: w_k = w0 / |G_k|^2 (= norm^2 |H_k|^2 / N0 for zero-forcing taps), one correctly
                            // rounded division per bin (uniform branch).  Only where the dispatch can ask for it (the
                            // instantiation without weight loads and the every-output ones): in the LLR-only kernel WITH a
                            // weight array the branch, never taken, split every bin's code and cost 1 to 2.5 points.
                            if constexpr (NOWGT || FULL) if (p.w_from_taps) w = p.w0 / (gk.x * gk.x + gk.y * gk.y);
                        }
                        if constexpr (!NOWGT) {
                            if (WGT_SURE || p.wgt) {
                                if constexpr (PREFETCH_TAPS) w = wv[h * GO + g];
                                else w = epi_w[g];
                            }
                        }
                    }
                    idx[g] = demap_point<float, MI, MQ, !HARDONLY, HARD_PATH || LC>(y.x, y.y, p.inorm, w, l + g * BPS);
                });
                if constexpr (STAGE) {
                    // SPW == 1: the wave's block is one contiguous span of the symbol.  SPW > 1 (one
                    // wave = SPW symbols): one contiguous run of TPS * CHUNK floats per symbol, so the
                    // copy is wave-wide and checks the destination symbol, not the lane's own.
                    if (want_llr() && (SPW > 1 || active)) {
                        // a strip without padding is linear memory: 16-byte vectors may straddle lanes
                        constexpr bool LINEAR = STG_STRIDE == CHUNK * 4 && (64 * CHUNK) % 4 == 0;
                        constexpr int VW = (CHUNK % 4 == 0 || LINEAR) ? 4 : 2;   // floats per vector
                        constexpr int NB = S::STG_BLOCKS;                        // blocks staged together (N = 64: two)
                        constexpr int hb = h % NB, h0 = h - hb;
                        constexpr int NV = 64 * CHUNK * NB / VW;                 // vectors per wave and staged group
                        constexpr int VPL = CHUNK / VW;                          // vectors per lane (padded strips)
                        const int lane = tid & 63;
                        float* stg = reinterpret_cast<float*>(lds + STG_OFF + (tid >> 6) * STG_WAVE_BYTES);
                        if constexpr (NB == 1) {
                            store_floats<CHUNK>(stg + lane * (STG_STRIDE / 4), l);
                        } else {
                            // symbol-major, the symbol's NB blocks one behind the other: memory order
                            static_assert(LINEAR && SPW > 1 && O::reg_pos(LAST, (h0 + NB - 1) * GO) == O::reg_pos(LAST, h0 * GO) + (NB - 1) * TPS * GO
                                              && O::reg_pos(LAST, (h0 + 1) * GO) == O::reg_pos(LAST, h0 * GO) + TPS * GO, "staged blocks are adjacent in memory");
                            store_floats<CHUNK>(stg + (((lane / TPS) * NB + hb) * TPS + (lane % TPS)) * CHUNK, l);
                        }
                        if constexpr (hb == NB - 1) {
                        wave_lds_sync();      // one wave: LDS serves its instructions in order, the fences keep the compiler from reordering them
                        constexpr int gbits = O::kLogE - P::kR[LAST];
                        float* gbase = p.llr + (bin0 + O::reg_pos(LAST, h0 * GO) + ((tid & ~63) << gbits)) * BPS;
                        static_for<(NV + 63) / 64>([&](auto I) {
                            constexpr int i = I;
                            const int v = i * 64 + lane;
                            const float* src;
                            if constexpr (LINEAR) src = stg + v * VW;
                            else src = stg + (v / VPL) * (STG_STRIDE / 4) + (v % VPL) * VW;
                            float* dst;
                            bool ok = (NV % 64 == 0) || v < NV;
                            if constexpr (SPW == 1) {
                                dst = gbase + v * VW;
                            } else {
                                constexpr int RUN = TPS * CHUNK * NB;            // floats per symbol in this staged group
                                static_assert(RUN % VW == 0, "vectors do not straddle symbols");
                                const int s_ = (v * VW) / RUN, off = (v * VW) % RUN;
                                const unsigned long long dsym = grp * SPW + s_;
                                ok = ok && dsym < p.n_sym;
                                dst = p.llr + (static_cast<size_t>(dsym) * N + O::reg_pos(LAST, h0 * GO)) * BPS + off;
                            }
                            if (ok) {
                                if constexpr (VW == 4) {
                                    const float4 q4 = *reinterpret_cast<const float4*>(src);
                                    stream_store4(dst, q4.x, q4.y, q4.z, q4.w);
                                } else {
                                    const float2 q2 = *reinterpret_cast<const float2*>(src);
                                    stream_store2(dst, q2.x, q2.y);
                                }
                            }
                        });
                        wave_lds_sync();      // one wave: LDS serves its instructions in order, the fences keep the compiler from reordering them
                        }      // last block of the staged group
                    }
                } else if constexpr (!HARDONLY) {
                    if (want_llr() && active) store_floats<GO * BPS>(p.llr + (bin0 + k) * BPS, l);
                }
                if constexpr (HARD_PATH || LC) {
                    if constexpr (HARD_PATH) if (HARDONLY || p.hard) static_for<GO>([&](auto G) { constexpr int g = G; lds_idx[slot * N + k + g] = static_cast<idx_t>(idx[g]); });
                    if constexpr (FULL) if (want_counts() && active) {
                        // GO adjacent points: GO * BPS <= 20 reference bits starting at a multiple of
                        // GO * BPS bits, i.e. inside a four-byte window that starts at the bit's byte
                        static_assert(GO * BPS <= 25, "reference bits of one register group fit a four-byte window");
                        const unsigned bit = static_cast<unsigned>(k) * BPS;
                        const unsigned b0 = bit >> 3;
                        const unsigned char* rs = lds_ref + slot * REF_SYM_BYTES;
                        unsigned window = 0;
                        static_for<(GO * BPS + 7 + 7) / 8>([&](auto I) {
                            constexpr unsigned i = I;
                            const unsigned bi = (b0 + i < REF_SYM_BYTES) ? b0 + i : REF_SYM_BYTES - 1;
                            window |= static_cast<unsigned>(rs[bi]) << (24u - 8u * i);
                        });
                        const unsigned both = (window >> (32u - (bit & 7u) - GO * BPS)) & ((1u << (GO * BPS)) - 1u);
                        unsigned n_bit = 0, n_sym_err = 0, n_sign = 0;                 // <= 20 LLRs: no overflow
                        const unsigned n_abs = llr_abs_sum<GO * BPS>(l, 0u);           // <= 20 * 65535
                        static_for<GO>([&](auto G) {
                            constexpr int g = G;
                            const unsigned ref = (both >> ((GO - 1 - g) * BPS)) & ((1u << BPS) - 1u);
                            const unsigned diff = ref ^ idx[g];
                            n_bit += __popc(diff);
                            n_sym_err += (diff != 0);
                            unsigned negative = 0;      // bit BPS-1-b = (LLR b < 0): the same order as the reference bits
                            static_for<BPS>([&](auto B) {
                                constexpr int b = B;
                                const float lv = l[g * BPS + b];
                                negative = (negative << 1) | (lv < 0.0f ? 1u : 0u);
                            });
                            n_sign += __popc(negative ^ ref);
                        });
                        c_bit += n_bit; c_sym += n_sym_err; c_sign += n_sign; c_abs += n_abs;
                    }
                }
            });
            if constexpr (HARD_PATH) pack_hard();
        }
    };
    process(grp_begin);
    if constexpr (LOOP && S::RUN > 2) {
        for (unsigned long long g = grp_begin + 1; g < grp_end; ++g) {      // uniform over the workgroup
            __syncthreads();
            process(g);
        }
    } else if constexpr (LOOP) {
        if (grp_begin + 1 < grp_end) {                 // uniform over the workgroup
            __syncthreads();                           // LDS regions are reused by the next symbol group
            process(grp_begin + 1);
        }
    }

    if constexpr (FULL && DEMAP) {
        if (want_counts()) {
            c_bit = wave_sum_u64(c_bit); c_sym = wave_sum_u64(c_sym); c_abs = wave_sum_u64(c_abs);
            c_sign = wave_sum_u64(c_sign);
            if ((tid & 63) == 0) {
                // p.counters is the context's slot array here (common.hpp: kCounterSlots x 8);
                // the host folds it into the caller's counters after the launch
                unsigned long long* dst = p.counters + static_cast<size_t>(blockIdx.x & (kCounterSlots - 1)) * 8;
                atomicAdd(&dst[0], c_bit);
                atomicAdd(&dst[2], c_sym);
                atomicAdd(&dst[5], c_abs);
                atomicAdd(&dst[6], c_sign);
            }
        }
    }
}

// Tuning knobs, read from the environment once per process (measurement aids; the defaults are
// what the benchmarks and tests run):
//   SDR_XCD_WINDOW     window of the XCD-aware workgroup-id permutation (0 = off)
//   SDR_LDS_PAD        bytes of unused dynamic LDS per workgroup (occupancy experiments)
//   SDR_DEBUG          print each instantiation's launch shape once
struct RxTuning {
    long xcd_window = 1L << 30;   // clipped to the grid: every XCD takes one contiguous eighth of the batch
    long lds_pad = 0;             // extra dynamic LDS per workgroup: throttles resident workgroups per CU
    long xcd_stagger = 0;         // SDR_XCD_STAGGER: XCD x starts x * this many workgroups into its share (phase of the eight streams)
    long chain_form = 0;          // SDR_CHAIN_FORM: the chunked link (link.hip) 0 = per shape, 1 = one kernel per chunk, 2 = generator + receive kernel
    bool debug = false;
    RxTuning()
    {
        if (const char* v = getenv("SDR_XCD_WINDOW")) xcd_window = atol(v) & ~7L;
        if (const char* v = getenv("SDR_LDS_PAD")) lds_pad = atol(v);
        if (const char* v = getenv("SDR_XCD_STAGGER")) xcd_stagger = atol(v) < 0 ? 0 : atol(v);
        if (const char* v = getenv("SDR_CHAIN_FORM")) chain_form = atol(v);
        debug = getenv("SDR_DEBUG") != nullptr;
    }
};
RxTuning& rx_tuning_mutable();      // rx_fused.hip
inline const RxTuning& rx_tuning() { return rx_tuning_mutable(); }

template <int LOG2N, int MI, int MQ, bool FULL, bool EQ, bool HARDONLY = false, bool NOWGT = false, bool LC = false>
int launch_rx_fused(sdr_ctx* ctx, const RxParams& p)
{
    using O = PlanOps<LOG2N>;
    using S = RxShape<LOG2N, MI, MQ, FULL, EQ, HARDONLY, NOWGT, LC>;
    auto kern = rx_fused_kernel<LOG2N, MI, MQ, FULL, EQ, HARDONLY, NOWGT, LC>;
    const RxTuning& tune = rx_tuning();
    const unsigned long long n_groups = (p.n_sym + O::kSPW - 1) / O::kSPW;
    // One symbol group per workgroup (with the XCD-aware order: +1.5 to +5 points over runs of two
    // at N = 512 ... 4096, and far better than a resident grid striding over the batch; the
    // per-workgroup twiddle set-up is cheap, the finer grain balances better).  The tap-holding
    // variants take two, reloading their taps once per run (65 % against 61 %).
    const unsigned long long per_wg = S::LOOP ? S::RUN : 1;
    if (tune.debug) {
        int nb = 0;
        hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, S::WG_THREADS, static_cast<size_t>(tune.lds_pad > 0 ? tune.lds_pad : 0));
        if (e != hipSuccess) return fail_hip(e, "hipOccupancyMaxActiveBlocksPerMultiprocessor");
        fprintf(stderr, "[sdr] rx_fused<%d,%d,%d,%d,%d,%d,%d,%d>: %d threads/WG, %d WG/CU, %d CUs\n", LOG2N, MI, MQ, int(FULL), int(EQ), int(HARDONLY), int(NOWGT), int(LC), S::WG_THREADS, nb, ctx->num_cu);
    }
    const unsigned long long blocks = (n_groups + per_wg - 1) / per_wg;
    if (blocks > 0x7FFFFFFFull) return fail(SDR_ERR_UNSUPPORTED, "batch too large for one launch");
    RxParams q = p;
    q.groups_per_wg = static_cast<uint32_t>(per_wg);
    // never wider than the grid: a batch smaller than the window would otherwise run unpermuted
    const unsigned long long win = tune.xcd_window > 0 ? static_cast<unsigned long long>(tune.xcd_window) : 0;
    q.xcd_window = static_cast<uint32_t>(win < (blocks & ~7ull) ? win : (blocks & ~7ull));
    q.xcd_stagger = (static_cast<unsigned long long>(tune.xcd_stagger) * 8 < (q.xcd_window >> 3)) ? static_cast<uint32_t>(tune.xcd_stagger) : 0u;
    hipLaunchKernelGGL(kern, dim3(static_cast<unsigned>(blocks)), dim3(S::WG_THREADS), static_cast<size_t>(tune.lds_pad > 0 ? tune.lds_pad : 0), ctx->stream, q);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail_hip(e, "rx_fused_kernel launch");
    return SDR_OK;
}

} // namespace sdr
