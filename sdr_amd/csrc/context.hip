// context.hip -- context lifetime, thread-local error text, twiddle-table cache, staging buffers.
#include "common.hpp"
#include "kernels.hpp"

#include <cmath>
#include <cstring>
#include <string>

namespace sdr {

static thread_local std::string g_last_error;

void set_error(const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
}

int fail(sdr_status code, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}

int fail_hip(hipError_t err, const char* what)
{
    g_last_error = std::string("HIP error: ") + hipGetErrorString(err) + " in " + what;
    return SDR_ERR_HIP;
}

int ensure_device(sdr_ctx* ctx)
{
    if (!ctx) return fail(SDR_ERR_INVALID, "null context");
    SDR_HIP(hipSetDevice(ctx->device));
    return SDR_OK;
}

// The reference's twiddle sequence (lib/inc/fft.hpp:113-150): per stage theta is formed in
// double and narrowed (:114); cos/sin in T (:115-116); w starts at 1+0i per block (:122-123),
// advances by one complex rotation written as four products (:140-142) and is renormalised
// by hypot when (j & 31) == 0 (:145-150).  Compiled with -ffp-contract=off, so each product
// and sum rounds separately, as in the reference's x86-64 build.
template <typename T>
static void make_twiddles(size_t n, bool inverse, T* out)
{
    size_t base = 0;
    for (size_t span = 2; span <= n; span *= 2) {
        const double direction = inverse ? -1.0 : 1.0;
        const T theta = static_cast<T>(-2.0 * direction * 3.14159265358979323846264338327950288 / static_cast<double>(span));
        const T rot_re = std::cos(theta);
        const T rot_im = std::sin(theta);
        const size_t count = span / 2;
        T w_re = static_cast<T>(1.0);
        T w_im = static_cast<T>(0.0);
        for (size_t j = 0; j < count; ++j) {
            out[2 * (base + j)] = w_re;
            out[2 * (base + j) + 1] = w_im;
            const T prev_re = w_re;
            w_re = prev_re * rot_re - w_im * rot_im;
            w_im = prev_re * rot_im + w_im * rot_re;
            if ((j & 31u) == 0) {
                const T len = std::hypot(w_re, w_im);
                w_re /= len;
                w_im /= len;
            }
        }
        base += count;
    }
}

void make_twiddles_f32(size_t n, bool inverse, float* out) { make_twiddles<float>(n, inverse, out); }
void make_twiddles_f64(size_t n, bool inverse, double* out) { make_twiddles<double>(n, inverse, out); }

int get_twiddles(sdr_ctx* ctx, size_t n, bool inverse, bool is_double, const void** d_table)
{
    const auto key = std::make_tuple(n, inverse ? 1 : 0, is_double ? 1 : 0);
    auto it = ctx->twiddles.find(key);
    if (it != ctx->twiddles.end()) { *d_table = it->second; return SDR_OK; }
    const size_t elem = is_double ? 2 * sizeof(double) : 2 * sizeof(float);
    const size_t bytes = (n - 1) * elem;
    std::vector<unsigned char> host(bytes);
    if (is_double) make_twiddles_f64(n, inverse, reinterpret_cast<double*>(host.data()));
    else make_twiddles_f32(n, inverse, reinterpret_cast<float*>(host.data()));
    void* d = nullptr;
    SDR_HIP(hipMalloc(&d, bytes));
    // synchronous copy: the host vector dies at return; table creation is a one-off per size
    hipError_t e = hipMemcpy(d, host.data(), bytes, hipMemcpyHostToDevice);
    if (e != hipSuccess) { (void)hipFree(d); return fail_hip(e, "hipMemcpy(twiddles)"); }
    ctx->twiddles[key] = d;
    *d_table = d;
    return SDR_OK;
}

// Tables of the Gaussian generator (rng.hpp): 32 pairs (r_i, L_i) and 256 pairs (sin, cos), every entry a double
// result rounded to float (a CPU restatement of the generator builds the same with the same libm).
void make_rng_tables_f32(float* out)
{
    for (int i = 0; i < 32; ++i) {
        const double c = 1.0 + (static_cast<double>(i) + 0.5) / 32.0;
        const float r = static_cast<float>(1.0 / c);
        out[2 * i] = r;
        out[2 * i + 1] = static_cast<float>(-std::log(static_cast<double>(r)));
    }
    for (int j = 0; j < 256; ++j) {
        const double a = 2.0 * 3.14159265358979323846264338327950288 * static_cast<double>(j) / 256.0;
        out[64 + 2 * j] = static_cast<float>(std::sin(a));
        out[64 + 2 * j + 1] = static_cast<float>(std::cos(a));
    }
}

int get_rng_tables(sdr_ctx* ctx, const float2** d_tables)
{
    if (!ctx->rng_tables) {
        float host[2 * (32 + 256)];
        make_rng_tables_f32(host);
        void* d = nullptr;
        SDR_HIP(hipMalloc(&d, sizeof(host)));
        hipError_t e = hipMemcpy(d, host, sizeof(host), hipMemcpyHostToDevice);      // synchronous, one-off per context
        if (e != hipSuccess) { (void)hipFree(d); return fail_hip(e, "hipMemcpy(rng tables)"); }
        ctx->rng_tables = d;
    }
    *d_tables = static_cast<const float2*>(ctx->rng_tables);
    return SDR_OK;
}

int get_staging(sdr_ctx* ctx, int slot, size_t bytes, void** ptr)
{
    Staging& s = ctx->stage[slot];
    if (s.bytes < bytes) {
        if (s.ptr) { SDR_HIP(hipStreamSynchronize(ctx->stream)); SDR_HIP(hipFree(s.ptr)); s.ptr = nullptr; s.bytes = 0; }
        const size_t want = bytes < 4096 ? 4096 : bytes;
        SDR_HIP(hipMalloc(&s.ptr, want));
        s.bytes = want;
    }
    *ptr = s.ptr;
    return SDR_OK;
}

int upload_table(sdr_ctx* ctx, const void* src, size_t bytes, int slot, void** d_ptr)
{
    if (!ctx->pinned_done) SDR_HIP(hipEventCreateWithFlags(&ctx->pinned_done, hipEventDisableTiming));
    else SDR_HIP(hipEventSynchronize(ctx->pinned_done));          // the previous table has left the pinned buffer
    if (ctx->pinned_bytes < bytes) {
        if (ctx->pinned) { SDR_HIP(hipHostFree(ctx->pinned)); ctx->pinned = nullptr; ctx->pinned_bytes = 0; }
        const size_t want = bytes < 16384 ? 16384 : bytes;
        SDR_HIP(hipHostMalloc(&ctx->pinned, want, hipHostMallocDefault));
        ctx->pinned_bytes = want;
    }
    memcpy(ctx->pinned, src, bytes);
    SDR_TRY(get_staging(ctx, slot, bytes, d_ptr));
    SDR_HIP(hipMemcpyAsync(*d_ptr, ctx->pinned, bytes, hipMemcpyHostToDevice, ctx->stream));
    SDR_HIP(hipEventRecord(ctx->pinned_done, ctx->stream));
    return SDR_OK;
}

int get_counter_slots(sdr_ctx* ctx, unsigned long long** slots)
{
    if (!ctx->counter_slots) {
        const size_t bytes = static_cast<size_t>(kCounterSlots) * 8 * sizeof(unsigned long long);
        SDR_HIP(hipMalloc(reinterpret_cast<void**>(&ctx->counter_slots), bytes));
        SDR_HIP(hipMemsetAsync(ctx->counter_slots, 0, bytes, ctx->stream));
    }
    *slots = ctx->counter_slots;
    return SDR_OK;
}

namespace {

// One workgroup of kCounterSlots threads: thread t owns slot t.  Device-scope atomics on ONE
// address from half a million workgroups serialise in the memory-side cache (measured: the
// counting receive path ran at an eighth of its speed); spread over 1024 lines they do not.
__global__ void __launch_bounds__(kCounterSlots) fold_counters_kernel(unsigned long long* slots, unsigned long long* out,
                                                                      unsigned long long n_points, unsigned bps, unsigned long long n_ofdm)
{
    __shared__ unsigned long long part[kCounterSlots / 64][4];
    const int t = threadIdx.x;
    unsigned long long v[4];
    constexpr int which[4] = {0, 2, 5, 6};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        v[j] = slots[t * 8 + which[j]];
        slots[t * 8 + which[j]] = 0;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const unsigned lo = __shfl_xor(static_cast<unsigned>(v[j]), off, 64);
            const unsigned hi = __shfl_xor(static_cast<unsigned>(v[j] >> 32), off, 64);
            v[j] += (static_cast<unsigned long long>(hi) << 32) | lo;
        }
        if ((t & 63) == 0) part[t >> 6][j] = v[j];
    }
    __syncthreads();
    if (t < 4) {
        unsigned long long s = 0;
        for (int w = 0; w < kCounterSlots / 64; ++w) s += part[w][t];
        atomicAdd(&out[which[t]], s);
    } else if (t == 4) {
        atomicAdd(&out[1], n_points * bps);
    } else if (t == 5) {
        atomicAdd(&out[3], n_points);
    } else if (t == 6) {
        atomicAdd(&out[4], n_ofdm);
    }
}

} // namespace

int fold_counters(sdr_ctx* ctx, unsigned long long* d_counters, uint64_t n_points, uint32_t bps, uint64_t n_ofdm)
{
    unsigned long long* slots = nullptr;
    SDR_TRY(get_counter_slots(ctx, &slots));
    return fold_counter_slots(ctx, slots, d_counters, n_points, bps, n_ofdm);
}

int fold_counter_slots(sdr_ctx* ctx, unsigned long long* slots, unsigned long long* d_counters, uint64_t n_points, uint32_t bps, uint64_t n_ofdm)
{
    hipLaunchKernelGGL(fold_counters_kernel, dim3(1), dim3(kCounterSlots), 0, ctx->stream, slots, d_counters,
                       static_cast<unsigned long long>(n_points), bps, static_cast<unsigned long long>(n_ofdm));
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail_hip(e, "fold_counters_kernel launch");
    return SDR_OK;
}

} // namespace sdr

using namespace sdr;

extern "C" {

const char* sdr_version(void) { return "sdr_amd 0.1 (gfx950)"; }
const char* sdr_last_error(void) { return g_last_error.c_str(); }

int sdr_device_count(int* count)
{
    if (!count) return fail(SDR_ERR_INVALID, "null count");
    SDR_HIP(hipGetDeviceCount(count));
    return SDR_OK;
}

static int create_common(int device, hipStream_t borrowed, bool borrow, sdr_ctx** out)
{
    if (!out) return fail(SDR_ERR_INVALID, "null out pointer");
    *out = nullptr;
    int n = 0;
    SDR_HIP(hipGetDeviceCount(&n));
    if (n <= 0) return fail(SDR_ERR_HIP, "no HIP device visible; libsdr_amd has no CPU fallback");
    if (device < 0 || device >= n) return fail(SDR_ERR_INVALID, "device %d out of range (0..%d)", device, n - 1);
    SDR_HIP(hipSetDevice(device));
    sdr_ctx* ctx = new sdr_ctx();
    ctx->device = device;
    if (borrow) {
        ctx->stream = borrowed;
        ctx->own_stream = false;
    } else {
        hipError_t e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
        if (e != hipSuccess) { delete ctx; return fail_hip(e, "hipStreamCreateWithFlags"); }
        ctx->own_stream = true;
    }
    hipDeviceProp_t prop;
    hipError_t e = hipGetDeviceProperties(&prop, device);
    if (e != hipSuccess) { sdr_destroy(ctx); return fail_hip(e, "hipGetDeviceProperties"); }
    ctx->num_cu = prop.multiProcessorCount;
    ctx->lds_max = prop.sharedMemPerBlockOptin ? prop.sharedMemPerBlockOptin : prop.sharedMemPerBlock;
    if (ctx->lds_max < prop.sharedMemPerBlock) ctx->lds_max = prop.sharedMemPerBlock;
    // the allocation-free entry point's static array: address and tables resolved here, once per device
    // (a failure there is that entry point's to report, not this one's)
    noalloc_attach(ctx);
    *out = ctx;
    return SDR_OK;
}

int sdr_create(int device, sdr_ctx** out) { return create_common(device, nullptr, false, out); }
int sdr_create_on_stream(int device, void* hip_stream, sdr_ctx** out)
{
    return create_common(device, static_cast<hipStream_t>(hip_stream), true, out);
}

void sdr_destroy(sdr_ctx* ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream || !ctx->own_stream) (void)hipStreamSynchronize(ctx->stream);
    for (auto& kv : ctx->twiddles) (void)hipFree(kv.second);
    for (auto& s : ctx->stage) if (s.ptr) (void)hipFree(s.ptr);
    if (ctx->counter_slots) (void)hipFree(ctx->counter_slots);
    if (ctx->rng_tables) (void)hipFree(ctx->rng_tables);
    noalloc_detach(ctx);
    if (ctx->aux_stream) { (void)hipStreamSynchronize(ctx->aux_stream); (void)hipStreamDestroy(ctx->aux_stream); }
    for (int i = 0; i < 2; ++i) {
        if (ctx->chain_generated[i]) (void)hipEventDestroy(ctx->chain_generated[i]);
        if (ctx->chain_consumed[i]) (void)hipEventDestroy(ctx->chain_consumed[i]);
    }
    if (ctx->pinned) (void)hipHostFree(ctx->pinned);
    if (ctx->pinned_done) (void)hipEventDestroy(ctx->pinned_done);
    if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

int sdr_synchronize(sdr_ctx* ctx)
{
    SDR_TRY(ensure_device(ctx));
    SDR_HIP(hipStreamSynchronize(ctx->stream));
    return SDR_OK;
}

void* sdr_stream(sdr_ctx* ctx) { return ctx ? static_cast<void*>(ctx->stream) : nullptr; }

int sdr_twiddle_table_c32(size_t n, int inverse, float* h_out)
{
    if (!h_out) return fail(SDR_ERR_INVALID, "null output");
    if (n < 2) return SDR_OK;
    if (!is_pow2(n)) return fail(SDR_ERR_NOT_POW2, "%s", kPow2Message);
    make_twiddles_f32(n, inverse != 0, h_out);
    return SDR_OK;
}

int sdr_rng_tables_f32(float* h_out)
{
    if (!h_out) return fail(SDR_ERR_INVALID, "null output");
    make_rng_tables_f32(h_out);
    return SDR_OK;
}

int sdr_twiddle_table_c64(size_t n, int inverse, double* h_out)
{
    if (!h_out) return fail(SDR_ERR_INVALID, "null output");
    if (n < 2) return SDR_OK;
    if (!is_pow2(n)) return fail(SDR_ERR_NOT_POW2, "%s", kPow2Message);
    make_twiddles_f64(n, inverse != 0, h_out);
    return SDR_OK;
}

} // extern "C"
