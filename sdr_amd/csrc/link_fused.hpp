// link_fused.hpp -- tuned transmit-side and Monte-Carlo kernels for n_fft = 64..8192 (fp32).
//
//   tx_fused_kernel   ofdm::tx for a batch (lib/inc/ofdm.hpp:13-24: IFFT with the reference
//                     network, /N, cyclic prefix), optionally fed by the Philox index generator +
//                     to_constl (lib/inc/modulation.hpp:78-94) + channel, and optionally followed
//                     by AWGN: the device frame generator.
//   mc_fused_kernel   generate -> IFFT -> + noise -> FFT -> demap -> count, one symbol per
//                     workgroup pass, nothing but eight counters ever written to HBM
//                     (model: app/QFDemoWindow.cpp:201-224 with a channel inserted).
// Same plans, same exchange layouts and the same operation DAG as rx_fused.hpp; the inverse
// transform only swaps the twiddle table (theta sign, lib/inc/fft.hpp:114) and divides by N
// (:200-202).  Definitions of the generator (no reference counterpart) are in rng.hpp/DESIGN.md.
#pragma once

#include "rng.hpp"
#include <type_traits>

#include "rx_fused.hpp"

namespace sdr {

struct TxParams {
    uint32_t vec16;              // output rows allow 16-byte stores (even cp, aligned base)
    const float2* freq;          // n_sym x N bins (SRC_GLOBAL) or null
    const float2* tw_inv;        // N-1 inverse twiddles
    const float2* chan;          // n_sets x N or null (generator only)
    const float2* rng;           // tables of the Gaussian generator (NOISE only)
    float2* out;                 // n_sym x (N + cp)
    uint8_t* tx_bits;            // packed transmitted bits or null (generator only)
    unsigned long long n_sym;
    unsigned long long seed;
    unsigned long long first_symbol;
    uint32_t cp;
    uint32_t sym_per_set;
    float sigma;                 // 0 with NOISE = false
    float norm;
    uint32_t groups_per_wg;
    uint32_t skip_prefix;        // generator: the cyclic prefix is neither computed nor stored (a receiver never reads it, lib/inc/ofdm.hpp:41)
};

struct McParams {
    const float2* tw_fwd;
    const float2* tw_inv;
    const float2* rng;           // tables of the Gaussian generator
    unsigned long long* counters;
    unsigned long long n_sym;
    unsigned long long seed;
    unsigned long long first_symbol;
    float sigma;
    float w0;
    float norm;
    float inorm;
    uint32_t groups_per_wg;
};

__device__ __forceinline__ int slot_of(unsigned tid, int tps) { return static_cast<int>(tid) / tps; }

// integer grid level of Gray code g on an axis with M bits, as float
template <int M>
__device__ __forceinline__ float level_of(unsigned g)
{
    if constexpr (M == 0) return 0.0f;
    const unsigned b = g ^ (g >> 1) ^ (g >> 2) ^ (g >> 3) ^ (g >> 4);
    return static_cast<float>(2 * static_cast<int>(b & ((1u << M) - 1u)) - ((1 << M) - 1));
}

// Transmitted indices of the E elements a thread holds, sharing Philox blocks: element index
// i = IDX(m) | thread_part; block = i mod (N/8), 16-bit lane = i / (N/8) = the top three index
// bits, which are register bits in both the first-pass input pattern and the last-pass output
// pattern of every plan (their butterfly bits are the top >= 3 bits).
template <int LOG2N, bool OUTPUT_SIDE>
struct TxIndexMap {
    using O = PlanOps<LOG2N>;
    using A = FftAddr<LOG2N>;
    static constexpr int kStrideLog = LOG2N - 3;
    static constexpr int idx_of(int m) { return OUTPUT_SIDE ? A::out_reg(m) : A::in_reg(m); }
    static constexpr int lane_mask()   // register bits that carry index bits >= kStrideLog
    {
        int mk = 0;
        for (int i = 0; i < O::kLogE; ++i)
            if (idx_of(1 << i) >> kStrideLog) mk |= 1 << i;
        return mk;
    }
};

template <int LOG2N, int BPS, bool OUTPUT_SIDE, typename F>
__device__ __forceinline__ void for_each_tx_index(unsigned long long seed, unsigned long long g, int thread_part, F&& f)
{
    using O = PlanOps<LOG2N>;
    using T = TxIndexMap<LOG2N, OUTPUT_SIDE>;
    constexpr int E = O::kE;
    constexpr int lane_mask = T::lane_mask();
    static_assert(O::popcount(lane_mask) == 3, "the top three index bits must be register bits");
    constexpr int rest_mask = (E - 1) & ~lane_mask;
    static_for<E / 8>([&](auto J) {
        constexpr int mrest = O::pdep(J, rest_mask);
        const uint32_t block = static_cast<uint32_t>((T::idx_of(mrest) | thread_part) & ((1 << T::kStrideLog) - 1));
        const Philox4 r = tx_block(seed, g, block);
        static_for<8>([&](auto L) {
            constexpr int m = mrest | O::pdep(L, lane_mask);
            constexpr uint32_t lane = static_cast<uint32_t>(T::idx_of(m) >> T::kStrideLog);
            const uint32_t word = r.v[lane >> 1];
            f(std::integral_constant<int, m>{}, ((word >> (16u * (lane & 1u))) & 0xFFFFu) & ((1u << BPS) - 1u));
        });
    });
}

#ifndef SDR_TX_STRAIGHT
#define SDR_TX_STRAIGHT 1
#endif
// Compile-time shape of one transmit instantiation (see RxShape in rx_fused.hpp for the reasoning).
#ifndef SDR_TX_DIRECT
#define SDR_TX_DIRECT 1
#endif
template <int LOG2N, int MI, int MQ, bool SRC_GLOBAL, bool NOISE>
struct TxShape {
    using O = PlanOps<LOG2N>;
    static constexpr int N = O::kN, SPW = O::kSPW;
    static constexpr int DATA_BYTES = O::max_sym_bytes() * SPW;
    static constexpr int TW_BYTES = O::kMidCount * 8;
    using idx_t = std::conditional_t<(MI + MQ > 8), uint16_t, uint8_t>;
    static constexpr int IDX_BYTES = SRC_GLOBAL ? 0 : N * SPW * static_cast<int>(sizeof(idx_t));
    // the plain batched ofdm::tx stores straight from the registers of the last pass (prefix = the
    // tail stored a second time); the generator assembles the symbol in LDS (rolled noise loops)
    static constexpr bool DIRECT = SDR_TX_DIRECT && SRC_GLOBAL && !NOISE && SPW == 1;      // several symbols per wave: the LDS-assembled rows store better (N = 64: 5.2e9 against 4.5e9 symbols/s)
    // The assembled symbol reuses the transform's exchange region, dead after the last exchange read
    // (one more barrier in multi-wave workgroups): 18.2 -> 10.2 KiB at N = 1024, 15 instead of 8
    // workgroups per CU by LDS, so that the registers (three waves per SIMD) set the residency.
#ifndef SDR_TX_TIME_ALIAS
#define SDR_TX_TIME_ALIAS 1
#endif
    static constexpr bool TIME_ALIAS = SDR_TX_TIME_ALIAS && !DIRECT && N * 8 * SPW <= DATA_BYTES;
    static constexpr int TIME_OFF = TIME_ALIAS ? 0 : (DATA_BYTES + TW_BYTES + IDX_BYTES + 15) & ~15;
    static constexpr int LDS_CORE = ((TIME_ALIAS ? DATA_BYTES + TW_BYTES + IDX_BYTES : TIME_OFF + (DIRECT ? 0 : N * 8 * SPW)) + 15) & ~15;
    static constexpr int RNG_OFF = LDS_CORE;                      // tables of the Gaussian generator (rng.hpp), NOISE only
    static constexpr int LDS_BYTES = LDS_CORE + (NOISE ? kRngBytes : 0) + 16;
    // one symbol group per workgroup as straight-line code (batched ofdm::tx: +20 % at N = 4096; the
    // ALU-bound generator: +2 to +7 % over runs of two)
    static constexpr bool LOOP = !SDR_TX_STRAIGHT;
    // two waves per SIMD where LDS allows two workgroups (left alone the compiler takes 256 + a few
    // registers), and never more per SIMD than an even spread of what LDS lets the CU hold
    static constexpr int WAVES_PER_WG = O::kWG / 64;
    static constexpr int LDS_WAVES = (160 * 1024 / ((LDS_BYTES + 511) & ~511)) * WAVES_PER_WG;
    static constexpr int MIN_WAVES = LOG2N <= 12 ? 2 : 1;
    static constexpr int even_share() { const int w = (LDS_WAVES + 3) / 4; return w < 1 ? 1 : w > 8 ? 8 : w; }
#ifdef SDR_NO_MAX_WAVES
    static constexpr int MAX_WAVES = 8;
#else
    static constexpr int MAX_WAVES = even_share() < MIN_WAVES ? MIN_WAVES : even_share();
#endif
};

// MI + MQ == 0 with SRC_GLOBAL: plain batched ofdm::tx.  SRC_GLOBAL = false: Philox generator.
template <int LOG2N, int MI, int MQ, bool SRC_GLOBAL, bool NOISE>
__global__ void __launch_bounds__(PlanOps<LOG2N>::kWG)
    __attribute__((amdgpu_waves_per_eu(TxShape<LOG2N, MI, MQ, SRC_GLOBAL, NOISE>::MIN_WAVES, TxShape<LOG2N, MI, MQ, SRC_GLOBAL, NOISE>::MAX_WAVES)))
    tx_fused_kernel(const TxParams p)
{
    constexpr bool LOOP = TxShape<LOG2N, MI, MQ, SRC_GLOBAL, NOISE>::LOOP;
    using O = PlanOps<LOG2N>;
    using P = typename O::P;
    using A = FftAddr<LOG2N>;
    constexpr int N = O::kN, E = O::kE, TPS = O::kTPS, SPW = O::kSPW;
    constexpr int BPS = MI + MQ;
    constexpr int SYM_BYTES = O::max_sym_bytes();
    constexpr int DATA_BYTES = SYM_BYTES * SPW;
    constexpr int TW_BYTES = O::kMidCount * 8;
    using idx_t = std::conditional_t<(MI + MQ > 8), uint16_t, uint8_t>;
    constexpr int IDX_BYTES = SRC_GLOBAL ? 0 : N * SPW * static_cast<int>(sizeof(idx_t));
    using S = TxShape<LOG2N, MI, MQ, SRC_GLOBAL, NOISE>;
    constexpr int TIME_OFF = S::TIME_OFF;
    static_assert(S::DATA_BYTES == DATA_BYTES && S::IDX_BYTES == IDX_BYTES, "TxShape mirrors the kernel's LDS layout");
    constexpr bool DIRECT = S::DIRECT;
    __shared__ __attribute__((aligned(16))) unsigned char lds[S::LDS_BYTES];
    idx_t* lds_idx = reinterpret_cast<idx_t*>(lds + DATA_BYTES + TW_BYTES);
    float2* lds_time = reinterpret_cast<float2*>(lds + TIME_OFF) + slot_of(threadIdx.x, TPS) * N;   // this symbol's time-domain samples
    [[maybe_unused]] const float2* lds_rng = reinterpret_cast<const float2*>(lds + S::RNG_OFF);
    // (several barriers lie between this copy and the first Gaussian sample)
    if constexpr (NOISE) rng_tables_to_lds(p.rng, reinterpret_cast<float2*>(lds + S::RNG_OFF), threadIdx.x, O::kWG);

    const int tid = threadIdx.x;
    const int slot = tid / TPS;
    const int t = tid % TPS;
    A addr;
    addr.init(t, slot, SYM_BYTES);
    FftTwiddles<LOG2N> tw;
    tw.init(p.tw_inv, reinterpret_cast<float2*>(lds + DATA_BYTES), tid, addr);
    if constexpr (O::kMidCount > 0) __syncthreads();

    const unsigned long long n_groups = (p.n_sym + SPW - 1) / SPW;
    const unsigned long long grp_begin = static_cast<unsigned long long>(blockIdx.x) * p.groups_per_wg;
    const unsigned long long grp_end = (grp_begin + p.groups_per_wg < n_groups) ? grp_begin + p.groups_per_wg : n_groups;
    const size_t out_stride = static_cast<size_t>(N) + p.cp;
    const unsigned long long first_set = p.sym_per_set ? p.first_symbol / p.sym_per_set : 0;

    if (grp_begin >= grp_end) return;      // uniform over the workgroup
    unsigned long long grp = grp_begin;
    do {
        unsigned long long sym = grp * SPW + slot;
        const bool active = sym < p.n_sym;
        if (!active) sym = p.n_sym - 1;
        const unsigned long long g = p.first_symbol + sym;

        float2 x[E];
        if constexpr (SRC_GLOBAL) {
            const float2* src = p.freq + sym * N + addr.in_thread;
            constexpr int GI = O::kGroupIn;
            static_for<E / GI>([&](auto Q) {
                constexpr int q = Q;
                constexpr int idx0 = A::in_reg(q);
                if constexpr (GI == 1) {
                    x[q] = src[idx0];
                } else {
                    const float4 v = *reinterpret_cast<const float4*>(src + idx0);
                    x[q] = make_float2(v.x, v.y);
                    x[q | (1 << P::kR[0])] = make_float2(v.z, v.w);
                }
            });
        } else {
            const float2* chan = nullptr;
            if (p.chan) {
                const size_t set = p.sym_per_set ? static_cast<size_t>(g / p.sym_per_set - first_set) : 0;
                chan = p.chan + set * N;
            }
            // the channel coefficients of the thread's E bins, requested together ahead of the index blocks (a load under the
            // per-element test below sat in a basic block of its own: sixteen exposed memory latencies per thread, and the
            // chain's generator ran at 7.9e7 symbols/s where the same kernel without a channel does 1.2e8)
            float2 hv[E];
            if (chan) static_for<E>([&](auto M) { constexpr int m = M; hv[m] = chan[A::in_reg(m) | addr.in_thread]; });
            for_each_tx_index<LOG2N, BPS, false>(p.seed, g, addr.in_thread, [&](auto M, uint32_t idx) {
                constexpr int m = M;
                float xr = level_of<MI>(idx >> MQ) * p.norm;                 // table[idx] * norm (modulation.hpp:90-91)
                float xi = level_of<MQ>(idx & ((1u << MQ) - 1u)) * p.norm;
                const int i = A::in_reg(m) | addr.in_thread;
                if (chan) {
                    const float2 h = hv[m];
                    const float tr = xr * h.x - xi * h.y;
                    const float ti = xr * h.y + xi * h.x;
                    xr = tr; xi = ti;
                }
                x[m] = make_float2(xr, xi);
                if (p.tx_bits) lds_idx[slot * N + i] = static_cast<idx_t>(idx);
            });
        }

        fft_run<LOG2N>(x, lds, addr, tw);

        // ---- /N (fft.hpp:200-202), noise, store with cyclic prefix (ofdm.hpp:21) ----
        // ifft2 divides by N (fft.hpp:200-202); N is a power of two, so the product with the exact
        // reciprocal is the same correctly rounded value as the quotient, at a tenth of the cost
        constexpr float INV_LEN = 1.0f / static_cast<float>(N);
        float2* dst = p.out + sym * out_stride;
        const int cp = static_cast<int>(p.cp);
        // The symbol is assembled in LDS in natural order, so that the noise is a rolled loop
        // (unrolled, sixteen Philox + Box-Muller chains per thread took 250 + 160 registers) and
        // the stores are contiguous over the lanes.
        if constexpr (DIRECT) {
            constexpr int GO = O::kGroupOut;
            const int tail = N - cp;                                 // samples from here on are the prefix too
            if (active) {
                static_for<E / GO>([&](auto Hh) {
                    constexpr int h = Hh;
                    const int k = A::out_reg(h * GO) | addr.pos_last;
                    if constexpr (GO == 2) {
                        const float4 v = make_float4(x[h * 2].x * INV_LEN, x[h * 2].y * INV_LEN, x[h * 2 + 1].x * INV_LEN, x[h * 2 + 1].y * INV_LEN);
                        if (p.vec16) {
                            *reinterpret_cast<float4*>(dst + cp + k) = v;
                            if (k >= tail) *reinterpret_cast<float4*>(dst + (k - tail)) = v;
                        } else {
                            dst[cp + k] = make_float2(v.x, v.y);
                            dst[cp + k + 1] = make_float2(v.z, v.w);
                            if (k >= tail) dst[k - tail] = make_float2(v.x, v.y);
                            if (k + 1 >= tail) dst[k + 1 - tail] = make_float2(v.z, v.w);
                        }
                    } else {
                        static_for<GO>([&](auto G) {
                            constexpr int g = G;
                            const float2 v = make_float2(x[h * GO + g].x * INV_LEN, x[h * GO + g].y * INV_LEN);
                            dst[cp + k + g] = v;
                            if (k + g >= tail) dst[k + g - tail] = v;
                        });
                    }
                });
            }
        } else {
        if constexpr (S::TIME_ALIAS && O::kWG > 64) __syncthreads();      // every wave is past its last exchange read
        static_for<E>([&](auto M) {
            constexpr int m = M;
            lds_time[A::out_reg(m) | addr.pos_last] = make_float2(x[m].x * INV_LEN, x[m].y * INV_LEN);
        });
        __syncthreads();
        // cyclic prefix first (ofdm.hpp:21 copies the clean tail; its noise is a stream of its own,
        // samples c and c + ceil(cp/2) sharing one Philox block)
        if (active && !p.skip_prefix) {
            if constexpr (NOISE) {
                const int half = static_cast<int>(cp_half(p.cp));
#pragma unroll 1
                for (int c = t; c < half && c < cp; c += TPS) {
                    const Philox4 b = cp_block(p.seed, g, static_cast<uint32_t>(c));
                    const float2 z0 = gaussian_pair(b.v[0], b.v[1], p.sigma, lds_rng);
                    const float2 v0 = lds_time[N - cp + c];
                    dst[c] = make_float2(v0.x + z0.x, v0.y + z0.y);
                    if (c + half < cp) {
                        const float2 z1 = gaussian_pair(b.v[2], b.v[3], p.sigma, lds_rng);
                        const float2 v1 = lds_time[N - cp + c + half];
                        dst[c + half] = make_float2(v1.x + z1.x, v1.y + z1.y);
                    }
                }
            } else {
#pragma unroll 1
                for (int c = t; c < cp; c += TPS) dst[c] = lds_time[N - cp + c];
            }
        }
        if constexpr (NOISE) {
            __syncthreads();                                        // prefix read the clean tail
#pragma unroll 1
            for (int i = 0; i < E / 2; ++i) {                       // samples n and n + N/2 share one Philox block
                const int n = i * TPS + t;
                const Philox4 b = noise_block(p.seed, g, static_cast<uint32_t>(n));
                const float2 z0 = gaussian_pair(b.v[0], b.v[1], p.sigma, lds_rng);
                const float2 z1 = gaussian_pair(b.v[2], b.v[3], p.sigma, lds_rng);
                const float2 a0 = lds_time[n], a1 = lds_time[n + N / 2];
                lds_time[n] = make_float2(a0.x + z0.x, a0.y + z0.y);
                lds_time[n + N / 2] = make_float2(a1.x + z1.x, a1.y + z1.y);
            }
            __syncthreads();
        }
        if (active) {
            if (p.vec16) {                                           // even cp, 16-byte aligned output: two samples per lane
                const float4* src4 = reinterpret_cast<const float4*>(lds_time);
                float4* dst4 = reinterpret_cast<float4*>(dst + cp);
#pragma unroll
                for (int i = 0; i < E / 2; ++i) dst4[i * TPS + t] = src4[i * TPS + t];
            } else {
#pragma unroll
                for (int i = 0; i < E; ++i) dst[cp + i * TPS + t] = lds_time[i * TPS + t];
            }
        }
        }      // !DIRECT
        if constexpr (!SRC_GLOBAL) {
            if (p.tx_bits) {
                __syncthreads();
                if (active) pack_decisions<N, BPS, idx_t>(lds_idx + slot * N, p.tx_bits + static_cast<size_t>(sym) * N * BPS / 8, t, TPS);
            }
        }
        if constexpr (LOOP) __syncthreads();
    } while (LOOP && ++grp < grp_end);
}

#ifdef SDR_MC_WPE
#define SDR_MC_ATTR __attribute__((amdgpu_waves_per_eu(SDR_MC_WPE)))
#else
#define SDR_MC_ATTR
#endif

// One workgroup's share of one (modulation, noise level) point of a Monte-Carlo run.
struct McWork {
    unsigned long long* counters;      // the point's eight counters
    unsigned long long n_sym;          // symbols of the point handled by this launch
    unsigned long long seed;
    unsigned long long first_symbol;   // global index of the point's first symbol in this launch
    unsigned long long grp_begin, grp_end;   // symbol groups of the point this workgroup runs ...
    unsigned long long grp_stride;     // ... in steps of this many (1: a contiguous run)
    float sigma, w0, norm, inorm;
    bool count_ofdm;                   // this workgroup adds n_sym to counters[4]
    // FADED bodies only (the chunked link, link.hip): one channel realisation, tap and weight per subcarrier and set of
    // sym_per_set symbols; LLRs of the launch's symbols (n_sym x N x bps) or null
    const float2* chan;
    const float2* taps;
    const float* wgt;
    float* llr;
    unsigned long long first_set;      // set of the launch's first symbol: the arrays start there
    uint32_t sym_per_set;
};

template <int LOG2N, bool FADED = false>
struct McLds {
    using O = PlanOps<LOG2N>;
    static constexpr int DATA_BYTES = O::max_sym_bytes() * O::kSPW;
    static constexpr int TW_BYTES = O::kMidCount * 8;
    // The time-domain symbol (IFFT output + noise, then the received bins for the demapping loop)
    // lives in the transform's exchange region, which is dead whenever it is in use: 17.2 -> 9.4 KiB
    // at N = 1024, so that registers, not LDS, bound the resident waves.
#ifndef SDR_MC_TIME_ALIAS
#define SDR_MC_TIME_ALIAS 1
#endif
    static constexpr bool TIME_ALIAS = SDR_MC_TIME_ALIAS && O::kN * 8 * O::kSPW <= DATA_BYTES;
    static constexpr int TIME_BYTES = TIME_ALIAS ? 0 : O::kN * 8 * O::kSPW;
    static constexpr int TIME_OFF = TIME_ALIAS ? 0 : DATA_BYTES + TW_BYTES;
    static constexpr int RNG_OFF = (DATA_BYTES + TW_BYTES + TIME_BYTES + 15) & ~15;      // tables of the Gaussian generator (rng.hpp)
    static constexpr int WGT_OFF = RNG_OFF + kRngBytes;                                  // FADED: the symbol's LLR weights for the rolled demapping loop
    static constexpr int BYTES = WGT_OFF + (FADED ? O::kN * 4 * O::kSPW : 0) + 16;
};

// generate -> IFFT -> + noise -> FFT -> demap -> count for the symbol groups [grp_begin, grp_end) of
// one point.  The modulation is a template parameter; set-up that does not depend on it (addresses,
// twiddles) is the caller's, so that one kernel can switch between several bodies.
// HARD: hard decisions only (the reference's own receive chain ofdm::rx -> from_constl): bit and
// symbol errors are counted, no LLR is formed and counters 5 and 6 stay untouched.
// FADED: the chunked link's body (link.hip: link_chain).  Every subcarrier is multiplied by its set's channel coefficient
// before the inverse transform (as the generator does, tx_fused_kernel), by its equaliser tap behind the forward transform
// and demapped with its own LLR weight (as the receive kernel does, rx_fused.hpp), and the LLRs are written out: the
// samples themselves never exist outside LDS.  Same operations in the same order as generator + receive kernel.
template <int LOG2N, int MI, int MQ, bool HARD = false, bool FADED = false>
__device__ __forceinline__ void mc_body(const McWork& p, unsigned char* lds, const FftAddr<LOG2N>& addr, const FftTwiddles<LOG2N>& tw_fwd)
{
    using O = PlanOps<LOG2N>;
    using A = FftAddr<LOG2N>;
    using L = McLds<LOG2N, FADED>;
    constexpr int N = O::kN, E = O::kE, TPS = O::kTPS, SPW = O::kSPW;
    constexpr int BPS = MI + MQ;
    static_assert(!(FADED && HARD), "the chunked link forms LLRs");
    float2* lds_time = reinterpret_cast<float2*>(lds + L::TIME_OFF);
    const float2* lds_rng = reinterpret_cast<const float2*>(lds + L::RNG_OFF);      // filled by the kernel's set-up
    [[maybe_unused]] float* lds_wgt = reinterpret_cast<float*>(lds + L::WGT_OFF);
    // aliased time buffer: other waves of the workgroup may still be inside the transform's last exchange
    constexpr bool ALIAS_SYNC = L::TIME_ALIAS && O::kWG > 64;
    const int tid = threadIdx.x;
    const int slot = tid / TPS;
    const int t = tid % TPS;

    unsigned long long c_bit = 0, c_sym = 0, c_abs = 0, c_sign = 0, c_n = 0;
    for (unsigned long long grp = p.grp_begin; grp < p.grp_end; grp += p.grp_stride) {
        unsigned long long sym = grp * SPW + slot;
        const bool active = sym < p.n_sym;
        if (!active) sym = p.n_sym - 1;
        const unsigned long long g = p.first_symbol + sym;

        // ---- transmit: indices -> constellation [-> x channel] -> IFFT -> /N -> + noise ----
        float2 x[E];
        [[maybe_unused]] size_t set_off = 0;                       // this symbol's set in the channel / tap / weight arrays
        if constexpr (FADED) set_off = static_cast<size_t>(g / p.sym_per_set - p.first_set) * N;
        // FADED: the channel coefficients of the eight elements that share a Philox block are requested together at the block's
        // first element (a load under the per-element code would be one exposed latency each, tx_fused_kernel); all E held at once,
        // as the generator does, cost this kernel a wave per SIMD
        constexpr int lane_mask = TxIndexMap<LOG2N, false>::lane_mask();
        [[maybe_unused]] float2 hv[8];
        for_each_tx_index<LOG2N, BPS, false>(p.seed, g, addr.in_thread, [&](auto M, uint32_t idx) {
            constexpr int m = M;
            float xr = level_of<MI>(idx >> MQ) * p.norm;
            float xi = level_of<MQ>(idx & ((1u << MQ) - 1u)) * p.norm;
            if constexpr (FADED) {
                if constexpr ((m & lane_mask) == 0)
                    static_for<8>([&](auto Q) { constexpr int mm = m | O::pdep(Q, lane_mask); hv[Q] = p.chan[set_off + (A::in_reg(mm) | addr.in_thread)]; });
                constexpr int q = [] { for (int i = 0; i < 8; ++i) if (O::pdep(i, lane_mask) == (m & lane_mask)) return i; return 0; }();
                const float2 h = hv[q];
                const float tr = xr * h.x - xi * h.y;
                const float ti = xr * h.y + xi * h.x;
                xr = tr; xi = ti;
            }
            x[m] = make_float2(xr, xi);
        });
        fft_run<LOG2N, true>(x, lds, addr, tw_fwd);
        // ifft2 divides by N (fft.hpp:200-202); N is a power of two, so the product with the exact
        // reciprocal is the same correctly rounded value as the quotient, at a tenth of the cost
        constexpr float INV_LEN = 1.0f / static_cast<float>(N);
        if constexpr (ALIAS_SYNC) __syncthreads();
        static_for<E>([&](auto M) {
            constexpr int m = M;
            const int n = A::out_reg(m) | addr.pos_last;
            lds_time[slot * N + n] = make_float2(x[m].x * INV_LEN, x[m].y * INV_LEN);
        });
        __syncthreads();
        // Noise as a ROLLED loop over the symbol's N/2 Philox blocks (samples n and n + N/2 share
        // one).  Fully unrolled, the sixteen independent Philox + Box-Muller chains drove the
        // kernel to 256 + 160 registers and one wave per SIMD.
#pragma unroll 1
        for (int i = 0; i < E / 2; ++i) {
            const int n = i * TPS + t;
            const Philox4 b = noise_block(p.seed, g, static_cast<uint32_t>(n));
            const float2 z0 = gaussian_pair(b.v[0], b.v[1], p.sigma, lds_rng);
            const float2 z1 = gaussian_pair(b.v[2], b.v[3], p.sigma, lds_rng);
            float2* lo = lds_time + slot * N + n;
            float2* hi = lo + N / 2;
            const float2 a0 = *lo, a1 = *hi;
            *lo = make_float2(a0.x + z0.x, a0.y + z0.y);
            *hi = make_float2(a1.x + z1.x, a1.y + z1.y);
        }
        __syncthreads();
        // ---- receive: the CP would be dropped here (ofdm.hpp:41); its samples are never made ----
        static_for<E>([&](auto M) { constexpr int m = M; x[m] = lds_time[slot * N + (A::in_reg(m) | addr.in_thread)]; });
        if constexpr (ALIAS_SYNC) __syncthreads();
        fft_run<LOG2N>(x, lds, addr, tw_fwd);

        // Demapping and counting as ROLLED loops over LDS: thread t takes Philox block j*TPS + t,
        // i.e. the eight subcarriers block + l*(N/8).  Unrolled in registers, sixteen interleaved
        // demappers (2 x 2^(bps/2) distances each) needed 500+ registers at 64/256-QAM.
        if constexpr (ALIAS_SYNC) __syncthreads();
        if constexpr (FADED) {
            // one-tap equaliser, the receive kernel's four products (rx_fused.hpp); taps and weights are requested behind the
            // transform, eight bins at a time: held across it they cost the kernel half its waves (218 registers against 127)
            constexpr int TG = E < 8 ? E : 8;
            static_for<E / TG>([&](auto Q) {
                constexpr int q = Q;
                float2 gv[TG];
                float wv[TG];
                // (the empty asm pins the requests here: with the address known from the symbol's start the scheduler moved them ahead
                // of the transform and held 48 registers across it)
                int pos = addr.pos_last;
                asm volatile("" : "+v"(pos));
                static_for<TG>([&](auto I) {
                    constexpr int m = q * TG + I;
                    const size_t k = set_off + (A::out_reg(m) | pos);
                    gv[I] = p.taps[k];
                    wv[I] = p.wgt[k];
                });
                static_for<TG>([&](auto I) {
                    constexpr int m = q * TG + I;
                    const int k = slot * N + (A::out_reg(m) | addr.pos_last);
                    const float er = x[m].x * gv[I].x - x[m].y * gv[I].y;
                    const float ei = x[m].x * gv[I].y + x[m].y * gv[I].x;
                    lds_time[k] = make_float2(er, ei);
                    lds_wgt[k] = wv[I];
                });
            });
        } else {
            static_for<E>([&](auto M) { constexpr int m = M; lds_time[slot * N + (A::out_reg(m) | addr.pos_last)] = x[m]; });
        }
        __syncthreads();
        unsigned n_bit = 0, n_sym_err = 0, n_sign = 0, n_abs = 0;   // per symbol and thread: <= 16 points x 10 bits (x 65535 for n_abs)
#pragma unroll 1
        for (int j = 0; j < E / 8; ++j) {
            const int block = j * TPS + t;
            const Philox4 r = tx_block(p.seed, g, static_cast<uint32_t>(block));
#pragma unroll 2      // two interleaved demappers: +1 to +4 % (stand-alone bodies), sweep +1.3 %; four: the sweep kernel (all six bodies) loses 12 % to registers
            for (int lane = 0; lane < 8; ++lane) {
                const int pair = lane >> 1;
                const uint32_t word = pair == 0 ? r.v[0] : pair == 1 ? r.v[1] : pair == 2 ? r.v[2] : r.v[3];
                const uint32_t ref = ((word >> (16u * (lane & 1u))) & 0xFFFFu) & ((1u << BPS) - 1u);
                const int bin = block + lane * (N / 8);
                const float2 y = lds_time[slot * N + bin];
                float l[BPS];
                float w = p.w0;
                if constexpr (FADED) w = lds_wgt[slot * N + bin];
                const unsigned idx = demap_point<float, MI, MQ, !HARD, true>(y.x, y.y, p.inorm, w, l);
                if constexpr (FADED) {
                    // BPS floats per lane, lane-contiguous over the TPS threads of the symbol, as streaming stores (+5 % over plain
                    // ones; a detour through LDS for 16 bytes per lane was level: the instruction shape is not what the stores cost)
                    if (p.llr && active) {
                        float* d_ = p.llr + (static_cast<size_t>(sym) * N + bin) * BPS;
                        if constexpr (BPS % 2 == 0) static_for<BPS / 2>([&](auto I) { constexpr int i = I; stream_store2(d_ + 2 * i, l[2 * i], l[2 * i + 1]); });
                        else d_[0] = l[0];
                    }
                }
                const unsigned diff = ref ^ idx;
                n_bit += __popc(diff);
                n_sym_err += (diff != 0);
                if constexpr (HARD) continue;
                unsigned negative = 0;
                static_for<BPS>([&](auto B) { constexpr int b = B; negative = (negative << 1) | (l[b] < 0.0f ? 1u : 0u); });
                n_sign += __popc(negative ^ ref);
                n_abs = llr_abs_sum<BPS>(l, n_abs);                 // counter 5 (qam.hpp); E * 10 * 65535 per symbol fits 32 bits
            }
        }
        if (active) { c_bit += n_bit; c_sym += n_sym_err; c_sign += n_sign; c_abs += n_abs; c_n += E; }
        __syncthreads();
    }

    c_bit = wave_sum_u64(c_bit); c_sym = wave_sum_u64(c_sym); c_abs = wave_sum_u64(c_abs);
    c_sign = wave_sum_u64(c_sign); c_n = wave_sum_u64(c_n);
    if ((tid & 63) == 0) {
        atomicAdd(&p.counters[0], c_bit);
        atomicAdd(&p.counters[1], c_n * BPS);
        atomicAdd(&p.counters[2], c_sym);
        atomicAdd(&p.counters[3], c_n);
        if constexpr (!HARD) {
            atomicAdd(&p.counters[5], c_abs);
            atomicAdd(&p.counters[6], c_sign);
        }
    }
    if (p.count_ofdm && tid == 0) atomicAdd(&p.counters[4], p.n_sym);
}

template <int LOG2N, int MI, int MQ>
__global__ void __launch_bounds__(PlanOps<LOG2N>::kWG) SDR_MC_ATTR mc_fused_kernel(const McParams p)
{
    using O = PlanOps<LOG2N>;
    constexpr int SPW = O::kSPW;
    __shared__ __attribute__((aligned(16))) unsigned char lds[McLds<LOG2N>::BYTES];
    const int tid = threadIdx.x;
    FftAddr<LOG2N> addr;
    addr.init(tid % O::kTPS, tid / O::kTPS, O::max_sym_bytes());
    // One twiddle set serves both directions: the inverse transform runs on the forward table
    // with conjugated butterflies (rx_fused.hpp, bfly<CONJ>).  That can differ from the true
    // inverse table only in the sign of exact zeros, which no counter can see; it halves the
    // twiddle registers of this register-bound kernel.
    FftTwiddles<LOG2N> tw_fwd;
    tw_fwd.init(p.tw_fwd, reinterpret_cast<float2*>(lds + McLds<LOG2N>::DATA_BYTES), tid, addr);
    rng_tables_to_lds(p.rng, reinterpret_cast<float2*>(lds + McLds<LOG2N>::RNG_OFF), tid, O::kWG);      // barriers follow before the first use
    if constexpr (O::kMidCount > 0) __syncthreads();

    const unsigned long long n_groups = (p.n_sym + SPW - 1) / SPW;
    McWork w;
    w.counters = p.counters;
    w.n_sym = p.n_sym; w.seed = p.seed; w.first_symbol = p.first_symbol;
    w.grp_begin = static_cast<unsigned long long>(blockIdx.x) * p.groups_per_wg;
    w.grp_end = (w.grp_begin + p.groups_per_wg < n_groups) ? w.grp_begin + p.groups_per_wg : n_groups;
    w.grp_stride = 1;
    w.sigma = p.sigma; w.w0 = p.w0; w.norm = p.norm; w.inorm = p.inorm;
    w.count_ofdm = blockIdx.x == 0;
    mc_body<LOG2N, MI, MQ>(w, lds, addr, tw_fwd);
}

// ---- the chunked faded link as ONE kernel per chunk ----------------------------------------------
struct ChainParams {
    const float2* tw_fwd;
    const float2* rng;
    const float2* chan;          // sets x N, first row = the set of first_symbol
    const float2* taps;
    const float* wgt;
    float* llr;                  // n_sym x N x bps or null
    unsigned long long* counters;
    unsigned long long n_sym;
    unsigned long long seed;
    unsigned long long first_symbol;
    float sigma, norm, inorm;
    uint32_t sym_per_set;
    uint32_t groups_per_wg;
};

// Registers: the Monte-Carlo body of the same shape fits four waves per SIMD with two registers to spare (126 of 128 at N = 2048);
// with the channel, tap and weight values in flight it takes 190.  Asked for three waves (168) it spills eight words and loses a
// tenth; two waves it is at N >= 2048, three below (no spills there).
template <int LOG2N, int MI>
constexpr int chain_min_waves() { return LOG2N <= 10 ? (MI == 5 ? 2 : 3) : LOG2N <= 12 ? 2 : 1; }

template <int LOG2N, int MI, int MQ>
__global__ void __launch_bounds__(PlanOps<LOG2N>::kWG) __attribute__((amdgpu_waves_per_eu(chain_min_waves<LOG2N, MI>(), 8))) chain_fused_kernel(const ChainParams p)
{
    using O = PlanOps<LOG2N>;
    using L = McLds<LOG2N, true>;
    constexpr int SPW = O::kSPW;
    __shared__ __attribute__((aligned(16))) unsigned char lds[L::BYTES];
    const int tid = threadIdx.x;
    FftAddr<LOG2N> addr;
    addr.init(tid % O::kTPS, tid / O::kTPS, O::max_sym_bytes());
    FftTwiddles<LOG2N> tw_fwd;                   // both directions from the forward table (mc_fused_kernel)
    tw_fwd.init(p.tw_fwd, reinterpret_cast<float2*>(lds + L::DATA_BYTES), tid, addr);
    rng_tables_to_lds(p.rng, reinterpret_cast<float2*>(lds + L::RNG_OFF), tid, O::kWG);      // barriers follow before the first use
    if constexpr (O::kMidCount > 0) __syncthreads();

    const unsigned long long n_groups = (p.n_sym + SPW - 1) / SPW;
    McWork w;
    w.counters = p.counters;
    w.n_sym = p.n_sym; w.seed = p.seed; w.first_symbol = p.first_symbol;
    // Workgroups are dealt round-robin over the 8 XCDs.  Each XCD takes one contiguous eighth of the symbol groups, and its
    // workgroups walk it side by side (workgroup i of the XCD: groups i, i + W, i + 2W, ...): what the resident workgroups write at
    // any moment is one compact window per XCD that moves forward, as in the receive kernel.  With a contiguous run per workgroup
    // (the Monte-Carlo kernel's split, nothing to write there) the LLRs went out as a thousand streams 6 MB apart and cost the
    // kernel a fifth of its time.
    const unsigned long long xcd = blockIdx.x & 7u, lane_wg = blockIdx.x >> 3, per_xcd = gridDim.x >> 3;      // the host launches a multiple of 8
    const unsigned long long share = (n_groups + 7) / 8;
    w.grp_begin = xcd * share + lane_wg;
    w.grp_end = ((xcd + 1) * share < n_groups) ? (xcd + 1) * share : n_groups;
    w.grp_stride = per_xcd;
    w.sigma = p.sigma; w.w0 = 0.0f; w.norm = p.norm; w.inorm = p.inorm;
    w.count_ofdm = blockIdx.x == 0;
    w.chan = p.chan; w.taps = p.taps; w.wgt = p.wgt; w.llr = p.llr;
    w.sym_per_set = p.sym_per_set;
    w.first_set = p.first_symbol / p.sym_per_set;
    mc_body<LOG2N, MI, MQ, false, true>(w, lds, addr, tw_fwd);
}

// ---- the whole Eb/N0 sweep in ONE launch --------------------------------------------------------
// Device-side description of one sweep point (host: link.hip builds the table, sorted so that the
// costliest modulations are dispatched first).  Workgroup b works on the point with the largest
// first_block <= b, on symbol groups [(b - first_block) * groups_per_block, ...).
struct SweepPointDev {
    unsigned long long first_symbol;
    unsigned long long n_sym;
    unsigned long long n_groups;
    unsigned long long first_block;
    float sigma, w0, norm, inorm;
    uint32_t mod;
    uint32_t row;                  // row of the caller's counter array
    uint32_t groups_per_block;
    uint32_t flags;                // sdr_sweep_point.flags (SDR_POINT_HARD_ONLY)
};
static_assert(sizeof(SweepPointDev) == 64, "one point per 64 bytes");

template <int LOG2N>
__global__ void __launch_bounds__(PlanOps<LOG2N>::kWG) SDR_MC_ATTR sweep_fused_kernel(const SweepPointDev* __restrict__ pts, uint32_t n_points, const float2* __restrict__ tw,
                                                                                      const float2* __restrict__ rng, unsigned long long seed, unsigned long long* __restrict__ counters)
{
    using O = PlanOps<LOG2N>;
    __shared__ __attribute__((aligned(16))) unsigned char lds[McLds<LOG2N>::BYTES];
    const int tid = threadIdx.x;
    FftAddr<LOG2N> addr;
    addr.init(tid % O::kTPS, tid / O::kTPS, O::max_sym_bytes());
    FftTwiddles<LOG2N> tw_fwd;
    tw_fwd.init(tw, reinterpret_cast<float2*>(lds + McLds<LOG2N>::DATA_BYTES), tid, addr);
    rng_tables_to_lds(rng, reinterpret_cast<float2*>(lds + McLds<LOG2N>::RNG_OFF), tid, O::kWG);      // barriers follow before the first use
    if constexpr (O::kMidCount > 0) __syncthreads();

    // uniform over the workgroup: the table is read through scalar loads
    const unsigned long long b = blockIdx.x;
    uint32_t lo = 0, hi = n_points;
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (pts[mid].first_block <= b) lo = mid; else hi = mid;
    }
    const SweepPointDev pt = pts[lo];
    const unsigned long long local = b - pt.first_block;
    McWork w;
    w.counters = counters + static_cast<size_t>(pt.row) * 8;
    w.n_sym = pt.n_sym; w.seed = seed; w.first_symbol = pt.first_symbol;
    w.grp_begin = local * pt.groups_per_block;
    w.grp_end = (w.grp_begin + pt.groups_per_block < pt.n_groups) ? w.grp_begin + pt.groups_per_block : pt.n_groups;
    w.grp_stride = 1;
    w.sigma = pt.sigma; w.w0 = pt.w0; w.norm = pt.norm; w.inorm = pt.inorm;
    w.count_ofdm = local == 0;
    if (w.grp_begin >= w.grp_end) return;          // uniform; the host never makes such a block
    if (pt.flags & 1u) {                           // SDR_POINT_HARD_ONLY
        switch (pt.mod) {
            case 1: mc_body<LOG2N, 1, 0, true>(w, lds, addr, tw_fwd); break;
            case 2: mc_body<LOG2N, 1, 1, true>(w, lds, addr, tw_fwd); break;
            case 4: mc_body<LOG2N, 2, 2, true>(w, lds, addr, tw_fwd); break;
            case 6: mc_body<LOG2N, 3, 3, true>(w, lds, addr, tw_fwd); break;
            case 8: mc_body<LOG2N, 4, 4, true>(w, lds, addr, tw_fwd); break;
            case 10: mc_body<LOG2N, 5, 5, true>(w, lds, addr, tw_fwd); break;
            default: break;
        }
        return;
    }
    switch (pt.mod) {
        case 1: mc_body<LOG2N, 1, 0>(w, lds, addr, tw_fwd); break;
        case 2: mc_body<LOG2N, 1, 1>(w, lds, addr, tw_fwd); break;
        case 4: mc_body<LOG2N, 2, 2>(w, lds, addr, tw_fwd); break;
        case 6: mc_body<LOG2N, 3, 3>(w, lds, addr, tw_fwd); break;
        case 8: mc_body<LOG2N, 4, 4>(w, lds, addr, tw_fwd); break;
        case 10: mc_body<LOG2N, 5, 5>(w, lds, addr, tw_fwd); break;
        default: break;                            // the host validates modulations
    }
}

inline unsigned long long resident_groups_per_wg(sdr_ctx* ctx, unsigned long long n_groups, int wg_per_cu)
{
    const unsigned long long resident = static_cast<unsigned long long>(ctx->num_cu) * static_cast<unsigned long long>(wg_per_cu > 0 ? wg_per_cu : 1);
    unsigned long long per = (n_groups + resident - 1) / resident;
    return per ? per : 1;
}

template <int LOG2N, int MI, int MQ, bool SRC_GLOBAL, bool NOISE>
int launch_tx_fused(sdr_ctx* ctx, const TxParams& p)
{
    using O = PlanOps<LOG2N>;
    const unsigned long long n_groups = (p.n_sym + O::kSPW - 1) / O::kSPW;
    unsigned long long per_wg = 2;
    static const long knob = [] { const char* v = getenv("SDR_TX_GROUPS_PER_WG"); return v ? atol(v) : 0L; }();   // tuning knob
    if (knob > 0) per_wg = static_cast<unsigned long long>(knob);
    if (!TxShape<LOG2N, MI, MQ, SRC_GLOBAL, NOISE>::LOOP) per_wg = 1;
    const unsigned long long blocks = (n_groups + per_wg - 1) / per_wg;
    if (blocks > 0x7FFFFFFFull) return fail(SDR_ERR_UNSUPPORTED, "batch too large for one launch");
    TxParams q = p;
    q.groups_per_wg = static_cast<uint32_t>(per_wg);
    q.vec16 = (p.cp % 2 == 0) && (reinterpret_cast<uintptr_t>(p.out) % 16 == 0);
    hipLaunchKernelGGL((tx_fused_kernel<LOG2N, MI, MQ, SRC_GLOBAL, NOISE>), dim3(static_cast<unsigned>(blocks)), dim3(O::kWG), 0, ctx->stream, q);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail_hip(e, "tx_fused_kernel launch");
    return SDR_OK;
}

template <int LOG2N>
int launch_sweep_fused(sdr_ctx* ctx, const SweepPointDev* d_pts, uint32_t n_points, unsigned long long blocks, const float2* tw, const float2* rng, unsigned long long seed, unsigned long long* counters)
{
    using O = PlanOps<LOG2N>;
    if (blocks == 0) return SDR_OK;
    if (blocks > 0x7FFFFFFFull) return fail(SDR_ERR_UNSUPPORTED, "sweep too large for one launch");
    hipLaunchKernelGGL((sweep_fused_kernel<LOG2N>), dim3(static_cast<unsigned>(blocks)), dim3(O::kWG), 0, ctx->stream, d_pts, n_points, tw, rng, seed, counters);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail_hip(e, "sweep_fused_kernel launch");
    return SDR_OK;
}

template <int LOG2N, int MI, int MQ>
int launch_chain_fused(sdr_ctx* ctx, const ChainParams& p)
{
    using O = PlanOps<LOG2N>;
    const unsigned long long n_groups = (p.n_sym + O::kSPW - 1) / O::kSPW;
    // a resident grid (atomics at the end of long runs, as launch_mc_fused), a multiple of 8 workgroups: see the kernel for the split
    unsigned long long per_wg = resident_groups_per_wg(ctx, n_groups, 8);
    if (per_wg < 4) per_wg = n_groups < 4 ? n_groups : 4;
    unsigned long long blocks = ((n_groups + per_wg - 1) / per_wg + 7) & ~7ull;
    ChainParams q = p;
    q.groups_per_wg = 0;
    hipLaunchKernelGGL((chain_fused_kernel<LOG2N, MI, MQ>), dim3(static_cast<unsigned>(blocks)), dim3(O::kWG), 0, ctx->stream, q);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail_hip(e, "chain_fused_kernel launch");
    return SDR_OK;
}

template <int LOG2N, int MI, int MQ>
int launch_mc_fused(sdr_ctx* ctx, const McParams& p)
{
    using O = PlanOps<LOG2N>;
    const unsigned long long n_groups = (p.n_sym + O::kSPW - 1) / O::kSPW;
    // compute bound and atomics at the end of every workgroup: a resident grid with long runs
    unsigned long long per_wg = resident_groups_per_wg(ctx, n_groups, 8);
    if (per_wg < 4) per_wg = n_groups < 4 ? n_groups : 4;
    const unsigned long long blocks = (n_groups + per_wg - 1) / per_wg;
    McParams q = p;
    q.groups_per_wg = static_cast<uint32_t>(per_wg > 0xFFFFFFFFull ? 0xFFFFFFFFull : per_wg);
    hipLaunchKernelGGL((mc_fused_kernel<LOG2N, MI, MQ>), dim3(static_cast<unsigned>(blocks)), dim3(O::kWG), 0, ctx->stream, q);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail_hip(e, "mc_fused_kernel launch");
    return SDR_OK;
}

} // namespace sdr
