// fft_generic.hip -- any-size, fp32 or fp64, batched radix-2 DIT FFT that evaluates the
// reference's network literally (lib/inc/fft.hpp:87-155): bit-reversal, then log2(n) stages of
// butterflies t = odd*w (four products), a' = even+t, b' = even-t, with w read from the table
// built by the reference's own recurrence (context.hip).  Compiled with -ffp-contract=off, so
// the result is the same IEEE operation DAG as the reference.
//
// Two forms:
//   * LDS form: one workgroup per sequence, the whole sequence resident in LDS (up to 128 KiB:
//     n <= 16384 c32 / 8192 c64), one barrier per stage.  HBM traffic = one read + one write.
//   * global form for longer sequences: one launch per stage over HBM.
// The tuned fp32 OFDM sizes (64..4096) never come here from the fused receive path; this is
// the generic engine behind fft::fft2/ifft2 and ofdm::rx/tx of the drop-in headers.
#include "common.hpp"
#include "kernels.hpp"

namespace sdr {

template <typename T> struct Vec2;
template <> struct Vec2<float> { using type = float2; };
template <> struct Vec2<double> { using type = double2; };

template <typename T>
struct FftGeom {
    size_t in_stride;     // elements between consecutive sequences on input
    size_t in_offset;     // elements to skip at the start of each input sequence (CP removal)
    size_t out_stride;
    size_t out_offset;    // elements left free at the start of each output sequence (CP slot)
    size_t cp_fill;       // tx: number of CP samples to copy from the tail (<= n)
    size_t n_seq;
    T divisor;            // inverse: divide every element by this (0 = no scaling)
};

template <typename T>
__device__ inline void butterfly(typename Vec2<T>::type& even, typename Vec2<T>::type& odd, typename Vec2<T>::type w)
{
    const T tr = odd.x * w.x - odd.y * w.y;
    const T ti = odd.x * w.y + odd.y * w.x;
    const T ar = even.x + tr, ai = even.y + ti;
    const T br = even.x - tr, bi = even.y - ti;
    even.x = ar; even.y = ai;
    odd.x = br; odd.y = bi;
}

// One sequence per `tps` threads, blockDim.x / tps sequences per workgroup (short sequences would
// otherwise leave most of a wave idle: the reference's own tests and demo use n = 8).
template <typename T>
__global__ void __launch_bounds__(1024) fft_lds_kernel(const typename Vec2<T>::type* __restrict__ in,
                                                      typename Vec2<T>::type* __restrict__ out,
                                                      const typename Vec2<T>::type* __restrict__ tw,
                                                      unsigned log2n, unsigned tps, FftGeom<T> g)
{
    using V = typename Vec2<T>::type;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const unsigned n = 1u << log2n;
    const unsigned spw = blockDim.x / tps;                 // sequences per workgroup
    const unsigned sub = threadIdx.x / tps;
    const unsigned tid = threadIdx.x - sub * tps;
    V* buf = reinterpret_cast<V*>(lds_raw) + static_cast<size_t>(sub) * n;
    const size_t n_groups = (g.n_seq + spw - 1) / spw;
    for (size_t grp = blockIdx.x; grp < n_groups; grp += gridDim.x) {
        const size_t seq = grp * spw + sub;
        const bool valid = seq < g.n_seq;                  // barriers below are reached by every thread
        if (valid) {
            const V* src = in + seq * g.in_stride + g.in_offset;
            // four loads requested together per trip (a thread usually owns exactly four points): one load, wait,
            // LDS store per trip left a single request in flight per thread
            const auto rev = [&](unsigned i_) { return log2n ? (__brev(i_) >> (32u - log2n)) : 0u; };
            unsigned i = tid;
            for (; i + 3u * tps < n; i += 4u * tps) {
                const V v0 = src[i], v1 = src[i + tps], v2 = src[i + 2u * tps], v3 = src[i + 3u * tps];
                buf[rev(i)] = v0;
                buf[rev(i + tps)] = v1;
                buf[rev(i + 2u * tps)] = v2;
                buf[rev(i + 3u * tps)] = v3;
            }
            for (; i < n; i += tps) buf[rev(i)] = src[i];
        }
        // Stages are taken two at a time: a thread holds the four positions that differ in bits c
        // and c+1 and runs the four literal radix-2 butterflies of both stages on them (same
        // operations and twiddles as one stage at a time; half the barriers and LDS round trips).
        unsigned c = 0;
        if (log2n & 1u) {                                  // odd number of stages: the first one alone
            __syncthreads();
            if (valid) {
                for (unsigned b = tid; b < (n >> 1); b += tps) {
                    V e = buf[2 * b], o = buf[2 * b + 1];
                    butterfly<T>(e, o, tw[0]);
                    buf[2 * b] = e;
                    buf[2 * b + 1] = o;
                }
            }
            c = 1;
        }
        for (; c + 1 < log2n; c += 2) {
            __syncthreads();
            const unsigned half = 1u << c;
            if (valid) {
                for (unsigned q = tid; q < (n >> 2); q += tps) {
                    const unsigned j = q & (half - 1u);
                    const unsigned p0 = ((q >> c) << (c + 2u)) | j;
                    V a0 = buf[p0], a1 = buf[p0 + half], a2 = buf[p0 + 2u * half], a3 = buf[p0 + 3u * half];
                    const V w0 = tw[half - 1u + j];                       // stage c, index j for both pairs
                    butterfly<T>(a0, a1, w0);
                    butterfly<T>(a2, a3, w0);
                    butterfly<T>(a0, a2, tw[2u * half - 1u + j]);         // stage c+1, index j
                    butterfly<T>(a1, a3, tw[2u * half - 1u + j + half]);  // stage c+1, index j + 2^c
                    buf[p0] = a0; buf[p0 + half] = a1; buf[p0 + 2u * half] = a2; buf[p0 + 3u * half] = a3;
                }
            }
        }
        __syncthreads();
        if (valid) {
            V* dst = out + seq * g.out_stride;
            for (unsigned i = tid; i < n; i += tps) {
                V v = buf[i];
                if (g.divisor != T(0)) { v.x /= g.divisor; v.y /= g.divisor; }
                dst[g.out_offset + i] = v;
                if (i + g.cp_fill >= n) dst[i + g.cp_fill - n] = v;   // cyclic prefix = copy of the tail
            }
        }
        __syncthreads();
    }
}

// ---- global-memory form -----------------------------------------------------------------
template <typename T>
__global__ void bitrev_gather_kernel(const typename Vec2<T>::type* __restrict__ in, typename Vec2<T>::type* __restrict__ out,
                                     unsigned log2n, FftGeom<T> g)
{
    const size_t n = size_t(1) << log2n;
    const size_t total = n * g.n_seq;
    for (size_t idx = blockIdx.x * size_t(blockDim.x) + threadIdx.x; idx < total; idx += size_t(gridDim.x) * blockDim.x) {
        const size_t seq = idx >> log2n;
        const size_t i = idx & (n - 1);
        const size_t r = log2n ? (__brevll(i) >> (64u - log2n)) : 0;
        out[seq * g.out_stride + g.out_offset + r] = in[seq * g.in_stride + g.in_offset + i];
    }
}

template <typename T>
__global__ void stage_global_kernel(typename Vec2<T>::type* __restrict__ data, const typename Vec2<T>::type* __restrict__ tw,
                                    unsigned log2n, unsigned c, FftGeom<T> g)
{
    using V = typename Vec2<T>::type;
    const size_t half_n = (size_t(1) << log2n) >> 1;
    const size_t total = half_n * g.n_seq;
    const size_t half = size_t(1) << c;
    for (size_t idx = blockIdx.x * size_t(blockDim.x) + threadIdx.x; idx < total; idx += size_t(gridDim.x) * blockDim.x) {
        const size_t seq = idx / half_n;
        const size_t b = idx - seq * half_n;
        const size_t j = b & (half - 1);
        const size_t pos = ((b >> c) << (c + 1)) | j;
        V* base = data + seq * g.out_stride + g.out_offset;
        V e = base[pos], o = base[pos + half];
        butterfly<T>(e, o, tw[half - 1 + j]);
        base[pos] = e;
        base[pos + half] = o;
    }
}

template <typename T>
__global__ void finish_global_kernel(typename Vec2<T>::type* __restrict__ data, unsigned log2n, FftGeom<T> g)
{
    using V = typename Vec2<T>::type;
    const size_t n = size_t(1) << log2n;
    const size_t total = n * g.n_seq;
    for (size_t idx = blockIdx.x * size_t(blockDim.x) + threadIdx.x; idx < total; idx += size_t(gridDim.x) * blockDim.x) {
        const size_t seq = idx >> log2n;
        const size_t i = idx & (n - 1);
        V* base = data + seq * g.out_stride;
        V v = base[g.out_offset + i];
        if (g.divisor != T(0)) { v.x /= g.divisor; v.y /= g.divisor; base[g.out_offset + i] = v; }
        if (i + g.cp_fill >= n) base[i + g.cp_fill - n] = v;
    }
}

template <typename T>
static int run_fft(sdr_ctx* ctx, const void* d_in, void* d_out, size_t n, bool inverse,
                   size_t in_stride, size_t in_offset, size_t out_stride, size_t out_offset, size_t cp_fill, size_t n_seq)
{
    using V = typename Vec2<T>::type;
    if (n_seq == 0 || n == 0) return SDR_OK;
    if (!is_pow2(n)) return fail(SDR_ERR_NOT_POW2, "%s", kPow2Message);
    if (!d_in || !d_out) return fail(SDR_ERR_INVALID, "null device pointer");
    if (n > (size_t(1) << 30)) return fail(SDR_ERR_UNSUPPORTED, "sequence length %zu too large", n);
    const unsigned log2n = ilog2(n);
    const void* tw = nullptr;
    if (n >= 2) SDR_TRY(get_twiddles(ctx, n, inverse, sizeof(T) == 8, &tw));
    FftGeom<T> g;
    g.in_stride = in_stride; g.in_offset = in_offset;
    g.out_stride = out_stride; g.out_offset = out_offset;
    g.cp_fill = cp_fill; g.n_seq = n_seq;
    g.divisor = inverse ? static_cast<T>(static_cast<ptrdiff_t>(n)) : T(0);
    const size_t lds_bytes = n * sizeof(V);
    const size_t lds_cap = ctx->lds_max < size_t(128) * 1024 ? ctx->lds_max : size_t(128) * 1024;
    const bool aliased = (d_in == d_out);
    if (lds_bytes <= lds_cap) {
        auto kern = fft_lds_kernel<T>;
        // threads per sequence: one per butterfly for short sequences; long ones (whose LDS footprint
        // leaves room for one or two workgroups per CU) get up to 1024 so that the CU is not left
        // with four waves
        unsigned tps = n / 2 < 256 ? static_cast<unsigned>(n / 2) : 256u;
        if (tps == 0) tps = 1;
        if (lds_bytes >= 32 * 1024) tps = n / 4 < 1024 ? static_cast<unsigned>(n / 4) : 1024u;
        unsigned threads = tps > 256 ? tps : 256;
        if (static_cast<size_t>(threads / tps) > n_seq) {                   // few sequences: a smaller workgroup
            threads = static_cast<unsigned>(n_seq) * tps;
            if (threads < 64) threads = 64 / tps * tps;
        }
        const unsigned spw = threads / tps;
        const size_t lds_total = lds_bytes * spw;
        if (lds_total > 48 * 1024)
            SDR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds_total)));
        size_t blocks = (n_seq + spw - 1) / spw;
        const size_t cap = size_t(ctx->num_cu) * 16;
        if (blocks > cap) blocks = cap;
        hipLaunchKernelGGL(kern, dim3(static_cast<unsigned>(blocks)), dim3(threads), lds_total, ctx->stream,
                           static_cast<const V*>(d_in), static_cast<V*>(d_out), static_cast<const V*>(tw), log2n, tps, g);
        SDR_HIP(hipGetLastError());
        return SDR_OK;
    }
    // global form; in-place needs a scratch copy for the bit-reversal gather
    const V* src = static_cast<const V*>(d_in);
    FftGeom<T> gg = g;
    if (aliased) {
        void* tmp = nullptr;
        SDR_TRY(get_staging(ctx, 2, n * n_seq * sizeof(V), &tmp));
        // compact copy of the inputs
        for (size_t s = 0; s < n_seq; ++s)
            SDR_HIP(hipMemcpyAsync(static_cast<V*>(tmp) + s * n, src + s * in_stride + in_offset, n * sizeof(V), hipMemcpyDeviceToDevice, ctx->stream));
        src = static_cast<const V*>(tmp);
        gg.in_stride = n; gg.in_offset = 0;
    }
    const unsigned threads = 256;
    const unsigned blocks = static_cast<unsigned>(ctx->num_cu) * 8;
    hipLaunchKernelGGL(bitrev_gather_kernel<T>, dim3(blocks), dim3(threads), 0, ctx->stream, src, static_cast<V*>(d_out), log2n, gg);
    SDR_HIP(hipGetLastError());
    for (unsigned c = 0; c < log2n; ++c) {
        hipLaunchKernelGGL(stage_global_kernel<T>, dim3(blocks), dim3(threads), 0, ctx->stream, static_cast<V*>(d_out), static_cast<const V*>(tw), log2n, c, gg);
        SDR_HIP(hipGetLastError());
    }
    if (inverse || cp_fill) {
        hipLaunchKernelGGL(finish_global_kernel<T>, dim3(blocks), dim3(threads), 0, ctx->stream, static_cast<V*>(d_out), log2n, gg);
        SDR_HIP(hipGetLastError());
    }
    return SDR_OK;
}

int fft_generic_f32(sdr_ctx* ctx, const void* d_in, void* d_out, size_t n, bool inverse,
                    size_t in_stride, size_t in_offset, size_t out_stride, size_t out_offset, size_t cp_fill, size_t n_seq)
{
    return run_fft<float>(ctx, d_in, d_out, n, inverse, in_stride, in_offset, out_stride, out_offset, cp_fill, n_seq);
}

int fft_generic_f64(sdr_ctx* ctx, const void* d_in, void* d_out, size_t n, bool inverse,
                    size_t in_stride, size_t in_offset, size_t out_stride, size_t out_offset, size_t cp_fill, size_t n_seq)
{
    return run_fft<double>(ctx, d_in, d_out, n, inverse, in_stride, in_offset, out_stride, out_offset, cp_fill, n_seq);
}

} // namespace sdr
