// demap.hip -- element-wise constellation kernels: to_constl / from_constl / nearest
// (reference lib/inc/modulation.hpp:49-112), soft demapping and the zero-forcing tap helper
// (no reference counterpart), and the generic-size back end of the fused receive path.
#include "common.hpp"
#include "kernels.hpp"
#include "bitpack.hpp"
#include "qam.hpp"

#include <cmath>

namespace sdr {

namespace {

// One wave exchanging data through LDS: the hardware serves a wave's LDS instructions in order, so no
// s_barrier is needed -- but the compiler must not move the loads above the other lanes' stores.  The bare
// wave barrier intrinsic carries no memory semantics; the wavefront-scope fences do.
__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}


constexpr unsigned kThreads = 256;
#ifndef SDR_ELEM_XCD
#define SDR_ELEM_XCD 1
#endif

inline unsigned grid_for(sdr_ctx* ctx, size_t work)
{
    size_t blocks = (work + kThreads - 1) / kThreads;
    const size_t cap = size_t(ctx->num_cu) * 16;
    if (blocks > cap) blocks = cap;
    if (blocks == 0) blocks = 1;
    return static_cast<unsigned>(blocks);
}

// Grid-stride range of this thread over n_items, XCD-aware: workgroups are dealt round-robin over
// the 8 XCDs (blockIdx % 8), so each XCD takes one contiguous eighth of the items and its
// workgroups stride through it -- every XCD's stores form one sequential stream (write-only
// streams: 5.7 -> 6.3 TB/s in tools/membench5).  Grids that are not a multiple of 8 keep the
// plain grid-stride order.
struct ItemRange { size_t first, last, step; };
__device__ inline ItemRange xcd_item_range(size_t n_items)
{
#if SDR_ELEM_XCD
    if ((gridDim.x & 7u) == 0) {
        const size_t per = (((n_items + 7) / 8) + blockDim.x - 1) / blockDim.x * blockDim.x;      // items per XCD, whole workgroups
        const size_t x = blockIdx.x & 7u, w = blockIdx.x >> 3, wgs = gridDim.x >> 3;
        const size_t end = (x + 1) * per < n_items ? (x + 1) * per : n_items;
        return {x * per + w * blockDim.x + threadIdx.x, end, wgs * blockDim.x};
    }
#endif
    return {blockIdx.x * size_t(blockDim.x) + threadIdx.x, n_items, size_t(gridDim.x) * blockDim.x};
}

// MSB-first bit field of `nbits` (<= 16) starting at stream bit `pos`; touches only the bytes
// that hold those bits.
__device__ inline unsigned stream_bits(const uint8_t* __restrict__ s, size_t pos, int nbits)
{
    const size_t first = pos >> 3;
    const size_t last = (pos + nbits - 1) >> 3;
    unsigned acc = 0;
    for (size_t b = first; b <= last; ++b) acc = (acc << 8) | s[b];
    const unsigned span_bits = static_cast<unsigned>(last - first + 1) * 8u;
    const unsigned shift = span_bits - static_cast<unsigned>(pos & 7) - static_cast<unsigned>(nbits);
    return (acc >> shift) & ((1u << nbits) - 1u);
}

// to_constl (lib/inc/modulation.hpp:78-94 for any order): a thread maps two consecutive symbols,
// so that a lane's store is 16 bytes (fp32) and a wave instruction writes 1 KiB contiguous;
// the output is written once and never re-read here: non-temporal.
template <typename T, int MI, int MQ>
__global__ void to_constl_kernel(const uint8_t* __restrict__ bytes, size_t n_sym, T norm, T* __restrict__ out, bool wide_ok)
{
    constexpr int BPS = MI + MQ;
    typedef T vec4 __attribute__((ext_vector_type(4)));
    const size_t n_pairs = (n_sym + 1) / 2;
    const ItemRange range = xcd_item_range(n_pairs);
    const size_t n_in = (n_sym * BPS + 7) / 8;      // bytes of the bit stream
    for (size_t pr = range.first; pr < range.last; pr += range.step) {
        T v[4] = {T(0), T(0), T(0), T(0)};
        // The pair's 2 * BPS bits start at bit 2 * pr * BPS and lie inside a four-byte window that starts at that bit's
        // byte ((pos & 7) + 2 * BPS <= 27): four unconditional byte loads (index clamped to the stream), all in flight
        // together, instead of a data-dependent loop of one to three loads per symbol.
        const size_t pos = 2 * pr * BPS;
        const size_t b0 = pos >> 3;
        unsigned window = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) window = (window << 8) | bytes[b0 + i < n_in ? b0 + i : n_in - 1];
        const unsigned both = (window >> (32u - static_cast<unsigned>(pos & 7) - 2u * BPS)) & ((1u << (2 * BPS)) - 1u);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const size_t k = 2 * pr + h;
            if (k >= n_sym) break;
            const unsigned idx = (both >> ((1 - h) * BPS)) & ((1u << BPS) - 1u);
            const unsigned gi = idx >> MQ, gq = idx & ((1u << MQ) - 1u);
            // level = 2 * gray_decode(g) - (L - 1): a small integer, exact in T (the table entry)
            const T li = MI == 0 ? T(0) : static_cast<T>(2 * static_cast<int>(gray_to_binary(gi)) - ((1 << MI) - 1));
            const T lq = MQ == 0 ? T(0) : static_cast<T>(2 * static_cast<int>(gray_to_binary(gq)) - ((1 << MQ) - 1));
            v[2 * h] = li * norm;          // table[idx] * norm: complex * real (modulation.hpp:90-91)
            v[2 * h + 1] = lq * norm;
        }
        if (wide_ok && 2 * pr + 1 < n_sym) {
            const vec4 q = {v[0], v[1], v[2], v[3]};
            __builtin_nontemporal_store(q, reinterpret_cast<vec4*>(out + 4 * pr));
        } else {
            out[4 * pr] = v[0];
            out[4 * pr + 1] = v[1];
            if (2 * pr + 1 < n_sym) { out[4 * pr + 2] = v[2]; out[4 * pr + 3] = v[3]; }
        }
    }
}

template <typename T, int MI, int MQ>
__global__ void nearest_kernel(const T* __restrict__ sym, size_t n_sym, T inorm, uint16_t* __restrict__ index, bool pair_ok)
{
    typedef T vec2 __attribute__((ext_vector_type(2)));
    const ItemRange range = xcd_item_range(n_sym);
    size_t k = range.first;
    if (pair_ok) {
        // four symbols per lane requested together: one load per trip leaves a wave one memory latency per symbol
        for (; k + 3 * range.step < range.last; k += 4 * range.step) {
            vec2 v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = *reinterpret_cast<const vec2*>(sym + 2 * (k + j * range.step));
#pragma unroll
            for (int j = 0; j < 4; ++j) index[k + j * range.step] = static_cast<uint16_t>(demap_point<T, MI, MQ, false, true>(v[j].x, v[j].y, inorm, T(0), nullptr));
        }
    }
    for (; k < range.last; k += range.step)
        index[k] = static_cast<uint16_t>(demap_point<T, MI, MQ, false, true>(sym[2 * k], sym[2 * k + 1], inorm, T(0), nullptr));
}

// One thread per output byte: gathers the bits of the symbols that overlap it.
__global__ void pack_kernel(const uint16_t* __restrict__ index, size_t n_sym, int bps, size_t n_bytes, uint8_t* __restrict__ out)
{
    for (size_t b = blockIdx.x * size_t(blockDim.x) + threadIdx.x; b < n_bytes; b += size_t(gridDim.x) * blockDim.x) {
        unsigned v = 0;
        for (int i = 0; i < 8; ++i) {
            const size_t pos = b * 8 + i;
            const size_t k = pos / bps;
            const int within = static_cast<int>(pos - k * bps);
            const unsigned bit = (k < n_sym) ? ((index[k] >> (bps - 1 - within)) & 1u) : 0u;
            v = (v << 1) | bit;
        }
        out[b] = static_cast<uint8_t>(v);
    }
}

// from_constl in one pass (lib/inc/modulation.hpp:96-112 for any order).  A wave takes 512
// consecutive symbols: lane-contiguous loads (1 KiB per instruction), hard decisions parked in a
// per-wave LDS strip, then lane l packs symbols 8l .. 8l+7 = BPS whole bytes of the MSB-first
// stream.  Nothing but the symbols is read and nothing but the bytes written (the two-kernel
// form went through a 2-byte index array in HBM).
template <typename T, int MI, int MQ>
__global__ void __launch_bounds__(256) from_constl_kernel(const T* __restrict__ sym, size_t n_sym, T inorm, size_t n_bytes, uint8_t* __restrict__ out, bool pair_ok, bool words_ok)
{
    constexpr int BPS = MI + MQ;
    typedef T vec2 __attribute__((ext_vector_type(2)));
    __shared__ __attribute__((aligned(16))) uint16_t strip[4][512];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint16_t* mine = strip[wave];
    const size_t n_tiles = (n_sym + 511) / 512;
    // natural tile order: this kernel is read-dominated, and the XCD-aware split that lifts the
    // write-heavy kernels above measured 4 % slower here
    for (size_t tile = static_cast<size_t>(blockIdx.x) * 4 + wave; tile < n_tiles; tile += static_cast<size_t>(gridDim.x) * 4) {
        const size_t k0 = tile * 512;
        if (pair_ok && k0 + 512 <= n_sym) {
            // whole tile: the eight loads are requested together, ahead of the decisions (under a per-symbol bounds
            // check every load sits in a basic block of its own and the wave pays eight memory latencies per tile)
            vec2 v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = *reinterpret_cast<const vec2*>(sym + 2 * (k0 + j * 64 + lane));
#pragma unroll
            for (int j = 0; j < 8; ++j) mine[j * 64 + lane] = static_cast<uint16_t>(demap_point<T, MI, MQ, false, true>(v[j].x, v[j].y, inorm, T(0), nullptr));
        } else
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const size_t k = k0 + j * 64 + lane;
            unsigned idx = 0;
            if (k < n_sym) {
                T re, im;
                if (pair_ok) { const vec2 v = *reinterpret_cast<const vec2*>(sym + 2 * k); re = v.x; im = v.y; }      // one load per symbol
                else { re = sym[2 * k]; im = sym[2 * k + 1]; }
                idx = demap_point<T, MI, MQ, false, true>(re, im, inorm, T(0), nullptr);
            }
            mine[j * 64 + lane] = static_cast<uint16_t>(idx);
        }
        wave_lds_sync();     // one wave: its LDS operations complete in order
        if (words_ok && k0 + 512 <= n_sym) {
            // A whole tile = 64 * BPS bytes = 16 * BPS four-byte words of the MSB-first stream: lane j builds word j from
            // the (at most CNT) symbols that overlap it, so that a store instruction writes 256 contiguous bytes.  (A lane
            // storing the BPS bytes of its own eight symbols one by one costs BPS instructions of 64 single bytes BPS
            // apart: more write requests than the kernel has read requests from 256-QAM up.)
            constexpr int WORDS = 16 * BPS;
            constexpr int CNT = (32 + BPS - 1) / BPS + 1;
            static_assert(CNT * BPS <= 64, "the symbols overlapping one word fit 64 bits");
            uint32_t* out32 = reinterpret_cast<uint32_t*>(out + k0 / 8 * BPS);
#pragma unroll
            for (int j0 = 0; j0 < WORDS; j0 += 64) {
                const int j = j0 + lane;
                if (j < WORDS) {
                    const int s0 = (32 * j) / BPS, off = 32 * j - s0 * BPS;
                    unsigned long long acc = 0;
#pragma unroll
                    for (int i = 0; i < CNT; ++i) acc = (acc << BPS) | mine[s0 + i < 512 ? s0 + i : 511];      // bits of symbols past the word are shifted out
                    out32[j] = __builtin_bswap32(static_cast<uint32_t>(acc >> (CNT * BPS - off - 32)));
                }
            }
            wave_lds_sync();
            continue;
        }
        unsigned long long lo = 0;           // bit stream of the group, MSB first: 8 * BPS <= 80 bits
        unsigned hi = 0;                     // bits above 64 (1024-QAM only)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const unsigned idx = mine[lane * 8 + i];
            if constexpr (8 * BPS > 64) hi = (hi << BPS) | static_cast<unsigned>(lo >> (64 - BPS));
            lo = (lo << BPS) | idx;
        }
        wave_lds_sync();
        const size_t first = (k0 / 8 + lane) * BPS;
#pragma unroll
        for (int b = 0; b < BPS; ++b) {
            const int shift = 8 * (BPS - 1 - b);         // byte b holds stream bits [8b, 8b+8)
            const unsigned v = shift >= 64 ? hi >> (shift - 64) : static_cast<unsigned>(lo >> shift);
            if (first + b < n_bytes) out[first + b] = static_cast<uint8_t>(v);
        }
    }
}

// from_constl for whole tiles of 1024 fp32 symbols (QPSK ... 256-QAM, 16-byte aligned input): a lane loads two adjacent
// symbols per instruction (1 KiB contiguous per wave instead of 512 bytes), parks both decisions with one 16-bit LDS
// store, and the wave packs the tile sixteen decisions per lane (bitpack.hpp: one 16-byte LDS read, BPS / 2 words, one
// store of 64 * 2 * BPS contiguous bytes per wave).  The ragged rest of an array goes through from_constl_kernel.
template <int MI, int MQ>
__global__ void __launch_bounds__(256) from_constl_wide_kernel(const float* __restrict__ sym, size_t n_tiles, float inorm, uint8_t* __restrict__ out)
{
    constexpr int BPS = MI + MQ;
    static_assert(BPS % 2 == 0 && BPS <= 8, "one-byte decisions of an even number of bits");
    __shared__ __attribute__((aligned(16))) uint8_t strip[4][1024];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint8_t* mine = strip[wave];
    for (size_t tile = static_cast<size_t>(blockIdx.x) * 4 + wave; tile < n_tiles; tile += static_cast<size_t>(gridDim.x) * 4) {
        const float4* src = reinterpret_cast<const float4*>(sym + 2 * tile * 1024);
        float4 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = src[j * 64 + lane];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const unsigned d0 = demap_point<float, MI, MQ, false, true>(v[j].x, v[j].y, inorm, 0.0f, nullptr);
            const unsigned d1 = demap_point<float, MI, MQ, false, true>(v[j].z, v[j].w, inorm, 0.0f, nullptr);
            *reinterpret_cast<uint16_t*>(mine + 2 * (j * 64 + lane)) = static_cast<uint16_t>(d0 | (d1 << 8));
        }
        wave_lds_sync();     // one wave: its LDS operations complete in order
        pack_decisions<1024, BPS, uint8_t>(mine, out + tile * (128 * BPS), lane, 64);
        wave_lds_sync();
    }
}

// Max-log LLRs of a symbol array.  From 6 bits per symbol up, a lane's LLRs (24-40 bytes) go through
// a per-wave LDS strip so that every store instruction writes 1 KiB contiguous (per-lane chunks cap a
// copy of this shape at 3 TB/s, tools/membench); the narrower orders store a lane's 4-16 bytes directly.
template <int MI, int MQ>
__global__ void __launch_bounds__(256) soft_demap_kernel(const float* __restrict__ sym, size_t n_sym, float inorm, const float* __restrict__ wgt, float w0,
                                                         float* __restrict__ llr, bool wide_ok)
{
    constexpr int BPS = MI + MQ;
    constexpr bool STAGED = BPS >= 6 && (64 * BPS) % 4 == 0;
    __shared__ __attribute__((aligned(16))) float strip[STAGED ? 4 : 1][STAGED ? 64 * BPS : 4];
    const ItemRange range = xcd_item_range(n_sym);
    const int lane = threadIdx.x & 63;
    // wave-uniform trip count: first - lane is the wave's first item
    // The next trip's symbol and weight are requested before this trip's LLRs are worked out (index clamped into the
    // range, so the loads are unconditional): two loads in flight per wave instead of one exposed latency per trip.
    struct In { float re, im, w; };
    auto fetch = [&](size_t k0_) {
        size_t k_ = k0_ + lane;
        if (k_ >= range.last) k_ = range.last - 1;
        return In{sym[2 * k_], sym[2 * k_ + 1], wgt ? wgt[k_] : w0};
    };
    In next = fetch(range.first - lane);
    for (size_t k0 = range.first - lane; k0 < range.last; k0 += range.step) {
        const size_t k = k0 + lane;
        const In cur = next;
        next = fetch(k0 + range.step < range.last ? k0 + range.step : k0);
        float l[BPS];
        if (k < range.last) demap_point<float, MI, MQ, true, false>(cur.re, cur.im, inorm, cur.w, l);
        if constexpr (STAGED) {
            if (wide_ok) {
                float* mine = strip[threadIdx.x >> 6];
                if (k < range.last) {
#pragma unroll
                    for (int b = 0; b < BPS; ++b) mine[lane * BPS + b] = l[b];
                }
                wave_lds_sync();     // one wave: its LDS operations complete in order
                const size_t n_here = (range.last - k0 < 64 ? range.last - k0 : 64) * BPS;      // floats of this wave's span
                float* dst = llr + k0 * BPS;
#pragma unroll
                for (int i = 0; i < BPS / 4 + (BPS % 4 != 0); ++i) {
                    const size_t f = (static_cast<size_t>(i) * 64 + lane) * 4;
                    if (f + 4 <= n_here) {
                        const float4 q = *reinterpret_cast<const float4*>(mine + f);
                        typedef float vec4 __attribute__((ext_vector_type(4)));
                        const vec4 v = {q.x, q.y, q.z, q.w};
                        __builtin_nontemporal_store(v, reinterpret_cast<vec4*>(dst + f));
                    } else {
                        for (size_t j = f; j < n_here && j < f + 4; ++j) dst[j] = mine[j];
                    }
                }
                wave_lds_sync();
                continue;
            }
        }
        if (k < range.last) {
#pragma unroll
            for (int b = 0; b < BPS; ++b) llr[k * BPS + b] = l[b];
        }
    }
}

__global__ void zf_taps_kernel(const float2* __restrict__ chan, size_t n, float norm2, float inv_n0, float2* __restrict__ taps, float* __restrict__ wgt)
{
    for (size_t k = blockIdx.x * size_t(blockDim.x) + threadIdx.x; k < n; k += size_t(gridDim.x) * blockDim.x) {
        const float2 h = chan[k];
        const float p = h.x * h.x + h.y * h.y;
        taps[k] = make_float2(h.x / p, -h.y / p);
        wgt[k] = (norm2 * p) * inv_n0;
    }
}

__device__ inline unsigned long long wave_sum(unsigned long long v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// Generic-size back end of the fused path: one thread per subcarrier of the whole batch.
template <int MI, int MQ>
__global__ void demap_fft_kernel(const float2* __restrict__ fft, uint32_t n_fft, uint64_t n_sym, uint32_t sym_per_set,
                                 const float2* __restrict__ eq, const float* __restrict__ wgt, float w0, bool w_from_taps, float inorm,
                                 float* __restrict__ llr, uint16_t* __restrict__ index,
                                 const uint8_t* __restrict__ ref_bits, unsigned long long* __restrict__ counters)
{
    constexpr int BPS = MI + MQ;
    const size_t total = size_t(n_sym) * n_fft;
    unsigned long long bit_err = 0, sym_err = 0, abs_q = 0, sign_err = 0, n_done = 0;
    for (size_t e = blockIdx.x * size_t(blockDim.x) + threadIdx.x; e < total; e += size_t(gridDim.x) * blockDim.x) {
        const size_t s = e / n_fft;
        const uint32_t k = static_cast<uint32_t>(e - s * n_fft);
        const size_t set = sym_per_set ? s / sym_per_set : 0;
        float2 y = fft[e];
        float w = wgt ? wgt[set * n_fft + k] : w0;
        if (eq) {
            const float2 g = eq[set * n_fft + k];
            const float er = y.x * g.x - y.y * g.y;
            const float ei = y.x * g.y + y.y * g.x;
            y.x = er; y.y = ei;
            if (w_from_taps) w = w0 / (g.x * g.x + g.y * g.y);      // SDR_DEMOD_WEIGHT_FROM_TAPS (no weight array)
        }
        float l[BPS];
        const unsigned idx = demap_point<float, MI, MQ, true, true>(y.x, y.y, inorm, w, l);
        if (llr) {
#pragma unroll
            for (int b = 0; b < BPS; ++b) llr[e * BPS + b] = l[b];
        }
        if (index) index[e] = static_cast<uint16_t>(idx);
        if (counters) {
            const unsigned ref = stream_bits(ref_bits, e * BPS, BPS);
            const unsigned diff = ref ^ idx;
            bit_err += __popc(diff);
            sym_err += (diff != 0);
            n_done += 1;
            abs_q += llr_abs_sum<BPS>(l, 0u);
#pragma unroll
            for (int b = 0; b < BPS; ++b) {
                const unsigned txbit = (ref >> (BPS - 1 - b)) & 1u;
                sign_err += (txbit != (l[b] < 0.0f ? 1u : 0u));
            }
        }
    }
    if (counters) {
        bit_err = wave_sum(bit_err); sym_err = wave_sum(sym_err); abs_q = wave_sum(abs_q); sign_err = wave_sum(sign_err); n_done = wave_sum(n_done);
        if ((threadIdx.x & 63) == 0) {
            atomicAdd(&counters[0], bit_err);
            atomicAdd(&counters[1], n_done * BPS);
            atomicAdd(&counters[2], sym_err);
            atomicAdd(&counters[3], n_done);
            atomicAdd(&counters[5], abs_q);
            atomicAdd(&counters[6], sign_err);
        }
    }
}

__global__ void add_counter_kernel(unsigned long long* counters, int slot, unsigned long long v) { atomicAdd(&counters[slot], v); }

template <typename T> T norm_in(int mod) { return T(1) / std::sqrt(static_cast<T>(grid_energy(mod))); }
template <typename T> T inorm_in(int mod) { return std::sqrt(static_cast<T>(grid_energy(mod))); }

} // namespace

int to_constl_dev(sdr_ctx* ctx, bool is_double, int mod, const uint8_t* d_bytes, size_t n_sym, void* d_out)
{
    if (!valid_mod(mod)) return fail(SDR_ERR_INVALID, "unknown modulation %d", mod);
    if (n_sym == 0) return SDR_OK;
    if (!d_bytes || !d_out) return fail(SDR_ERR_INVALID, "null device pointer");
    const unsigned grid = grid_for(ctx, (n_sym + 1) / 2);
    dispatch_mod(mod, [&]<int MI, int MQ>() {
        if (is_double)
            hipLaunchKernelGGL((to_constl_kernel<double, MI, MQ>), dim3(grid), dim3(kThreads), 0, ctx->stream, d_bytes, n_sym, norm_in<double>(mod), static_cast<double*>(d_out), reinterpret_cast<uintptr_t>(d_out) % 32 == 0);
        else
            hipLaunchKernelGGL((to_constl_kernel<float, MI, MQ>), dim3(grid), dim3(kThreads), 0, ctx->stream, d_bytes, n_sym, norm_in<float>(mod), static_cast<float*>(d_out), reinterpret_cast<uintptr_t>(d_out) % 16 == 0);
    });
    SDR_HIP(hipGetLastError());
    return SDR_OK;
}

static int launch_nearest(sdr_ctx* ctx, bool is_double, int mod, const void* d_sym, size_t n_sym, uint16_t* d_index)
{
    const unsigned grid = grid_for(ctx, n_sym);
    dispatch_mod(mod, [&]<int MI, int MQ>() {
        if (is_double)
            hipLaunchKernelGGL((nearest_kernel<double, MI, MQ>), dim3(grid), dim3(kThreads), 0, ctx->stream, static_cast<const double*>(d_sym), n_sym, inorm_in<double>(mod), d_index, reinterpret_cast<uintptr_t>(d_sym) % (2 * sizeof(double)) == 0);
        else
            hipLaunchKernelGGL((nearest_kernel<float, MI, MQ>), dim3(grid), dim3(kThreads), 0, ctx->stream, static_cast<const float*>(d_sym), n_sym, inorm_in<float>(mod), d_index, reinterpret_cast<uintptr_t>(d_sym) % (2 * sizeof(float)) == 0);
    });
    SDR_HIP(hipGetLastError());
    return SDR_OK;
}

int nearest_dev(sdr_ctx* ctx, bool is_double, int mod, const void* d_sym, size_t n_sym, uint16_t* d_index)
{
    if (!valid_mod(mod)) return fail(SDR_ERR_INVALID, "unknown modulation %d", mod);
    if (n_sym == 0) return SDR_OK;
    if (!d_sym || !d_index) return fail(SDR_ERR_INVALID, "null device pointer");
    return launch_nearest(ctx, is_double, mod, d_sym, n_sym, d_index);
}

int from_constl_dev(sdr_ctx* ctx, bool is_double, int mod, const void* d_sym, size_t n_sym, uint8_t* d_bytes)
{
    if (!valid_mod(mod)) return fail(SDR_ERR_INVALID, "unknown modulation %d", mod);
    const size_t n_bytes = (n_sym * static_cast<size_t>(mod)) / 8;
    if (n_bytes == 0) return SDR_OK;
    if (!d_sym || !d_bytes) return fail(SDR_ERR_INVALID, "null device pointer");
    static_assert(kThreads == 256, "from_constl_kernel is written for four waves per workgroup");
    // fp32, QPSK ... 256-QAM, aligned buffers: whole tiles of 1024 symbols through the wide kernel, the rest below
    if (!is_double && mod % 2 == 0 && mod <= 8 && n_sym >= 1024 && reinterpret_cast<uintptr_t>(d_sym) % 16 == 0 && reinterpret_cast<uintptr_t>(d_bytes) % 4 == 0) {
        const size_t n_tiles = n_sym / 1024;
        const unsigned wide_grid = grid_for(ctx, n_tiles * 64);
        dispatch_mod(mod, [&]<int MI, int MQ>() {
            if constexpr ((MI + MQ) % 2 == 0 && MI + MQ <= 8)
                hipLaunchKernelGGL((from_constl_wide_kernel<MI, MQ>), dim3(wide_grid), dim3(kThreads), 0, ctx->stream, static_cast<const float*>(d_sym), n_tiles, inorm_in<float>(mod), d_bytes);
        });
        SDR_HIP(hipGetLastError());
        const size_t done = n_tiles * 1024;
        if (done == n_sym) return SDR_OK;
        return from_constl_dev(ctx, false, mod, static_cast<const float*>(d_sym) + 2 * done, n_sym - done, d_bytes + done / 8 * static_cast<size_t>(mod));      // < 1024 symbols: the general kernel
    }
    const unsigned grid = grid_for(ctx, (n_sym + 7) / 8);
    dispatch_mod(mod, [&]<int MI, int MQ>() {
        if (is_double)
            hipLaunchKernelGGL((from_constl_kernel<double, MI, MQ>), dim3(grid), dim3(kThreads), 0, ctx->stream, static_cast<const double*>(d_sym), n_sym, inorm_in<double>(mod), n_bytes, d_bytes,
                               reinterpret_cast<uintptr_t>(d_sym) % (2 * sizeof(double)) == 0, reinterpret_cast<uintptr_t>(d_bytes) % 4 == 0);
        else
            hipLaunchKernelGGL((from_constl_kernel<float, MI, MQ>), dim3(grid), dim3(kThreads), 0, ctx->stream, static_cast<const float*>(d_sym), n_sym, inorm_in<float>(mod), n_bytes, d_bytes,
                               reinterpret_cast<uintptr_t>(d_sym) % (2 * sizeof(float)) == 0, reinterpret_cast<uintptr_t>(d_bytes) % 4 == 0);
    });
    SDR_HIP(hipGetLastError());
    return SDR_OK;
}

int soft_demap_dev(sdr_ctx* ctx, int mod, const void* d_sym, size_t n_sym, const float* d_weight, float weight0, float* d_llr)
{
    if (!valid_mod(mod)) return fail(SDR_ERR_INVALID, "unknown modulation %d", mod);
    if (n_sym == 0) return SDR_OK;
    if (!d_sym || !d_llr) return fail(SDR_ERR_INVALID, "null device pointer");
    const unsigned grid = grid_for(ctx, n_sym);
    dispatch_mod(mod, [&]<int MI, int MQ>() {
        hipLaunchKernelGGL((soft_demap_kernel<MI, MQ>), dim3(grid), dim3(kThreads), 0, ctx->stream, static_cast<const float*>(d_sym), n_sym, inorm_in<float>(mod), d_weight, weight0, d_llr,
                           reinterpret_cast<uintptr_t>(d_llr) % 16 == 0);
    });
    SDR_HIP(hipGetLastError());
    return SDR_OK;
}

int zf_taps_dev(sdr_ctx* ctx, int mod, const void* d_chan, size_t n, float inv_n0, void* d_taps, float* d_weight)
{
    if (!valid_mod(mod)) return fail(SDR_ERR_INVALID, "unknown modulation %d", mod);
    if (n == 0) return SDR_OK;
    if (!d_chan || !d_taps || !d_weight) return fail(SDR_ERR_INVALID, "null device pointer");
    const float nrm = norm_in<float>(mod);
    hipLaunchKernelGGL(zf_taps_kernel, dim3(grid_for(ctx, n)), dim3(kThreads), 0, ctx->stream, static_cast<const float2*>(d_chan), n, nrm * nrm, inv_n0,
                       static_cast<float2*>(d_taps), d_weight);
    SDR_HIP(hipGetLastError());
    return SDR_OK;
}

int demap_after_fft(sdr_ctx* ctx, const sdr_demod_args& a, const void* d_fft)
{
    const size_t total = size_t(a.n_sym) * a.n_fft;
    if (total == 0) return SDR_OK;
    const bool want_counters = a.d_counters && a.d_ref_hard;
    const bool want_index = a.d_hard != nullptr;
    if (!a.d_llr && !want_index && !want_counters) return SDR_OK;
    void* idx = nullptr;
    if (want_index) SDR_TRY(get_staging(ctx, 1, total * sizeof(uint16_t), &idx));
    const unsigned grid = grid_for(ctx, total);
    const int mod = static_cast<int>(a.mod);
    dispatch_mod(mod, [&]<int MI, int MQ>() {
        hipLaunchKernelGGL((demap_fft_kernel<MI, MQ>), dim3(grid), dim3(kThreads), 0, ctx->stream, static_cast<const float2*>(d_fft), a.n_fft, a.n_sym,
                           a.sym_per_set, static_cast<const float2*>(a.d_eq), a.d_llr_weight, a.llr_weight0,
                           (a.flags & SDR_DEMOD_WEIGHT_FROM_TAPS) != 0 && a.d_eq && !a.d_llr_weight, inorm_in<float>(mod), a.d_llr,
                           static_cast<uint16_t*>(idx), a.d_ref_hard, want_counters ? reinterpret_cast<unsigned long long*>(a.d_counters) : nullptr);
    });
    SDR_HIP(hipGetLastError());
    if (want_counters) {
        hipLaunchKernelGGL(add_counter_kernel, dim3(1), dim3(1), 0, ctx->stream, reinterpret_cast<unsigned long long*>(a.d_counters), 4, static_cast<unsigned long long>(a.n_sym));
        SDR_HIP(hipGetLastError());
    }
    if (want_index) {
        const size_t n_bytes = (total * a.mod) / 8;
        if (n_bytes) {
            hipLaunchKernelGGL(pack_kernel, dim3(grid_for(ctx, n_bytes)), dim3(kThreads), 0, ctx->stream, static_cast<const uint16_t*>(idx), total, mod, n_bytes, a.d_hard);
            SDR_HIP(hipGetLastError());
        }
    }
    return SDR_OK;
}

} // namespace sdr
