// membench4.hip -- read:write = 1:2 copies (the fused receive kernel's bytes at N=1024/16-QAM) in
// different temporal/spatial shapes: which arrangement of the same traffic does the memory system like?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float v4f __attribute__((ext_vector_type(4)));

constexpr int IN_ROWS = 8;     // 1 KiB rows read per symbol (8 KiB payload; the 2 KiB prefix is skipped)
constexpr int OUT_ROWS = 16;   // 1 KiB rows written per symbol
constexpr int IN_STRIDE_ROWS = 10;

__device__ __forceinline__ unsigned long long xcd_remap(unsigned long long wg, unsigned long long grid)
{
    const unsigned long long win = 4096, full = (grid / win) * win;
    if (wg < full) { const unsigned long long r = wg % win; wg = wg - r + (r & 7) * (win >> 3) + (r >> 3); }
    return wg;
}

// S0: one wave per symbol, all loads then all stores; WAVES symbols per workgroup (independent waves)
template <int WAVES, int REMAP>
__global__ void __launch_bounds__(64 * WAVES) s0(const v4f* __restrict__ in, v4f* __restrict__ out, unsigned long long n_sym)
{
    unsigned long long wg = REMAP ? xcd_remap(blockIdx.x, gridDim.x) : blockIdx.x;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int rep = 0; rep < 2; ++rep) {
        const unsigned long long sym = (wg * 2 + rep) * WAVES + w;
        if (sym >= n_sym) return;
        const v4f* src = in + (sym * IN_STRIDE_ROWS + 2) * 64;
        v4f acc = {0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < IN_ROWS; ++i) acc += src[i * 64 + lane];
        v4f* dst = out + sym * OUT_ROWS * 64;
#pragma unroll
        for (int i = 0; i < OUT_ROWS; ++i) __builtin_nontemporal_store(acc, dst + i * 64 + lane);
    }
}
// S1: WAVES waves share WAVES symbols: symbol s is read and written by all waves together, rows interleaved
template <int WAVES, int REMAP>
__global__ void __launch_bounds__(64 * WAVES) s1(const v4f* __restrict__ in, v4f* __restrict__ out, unsigned long long n_sym)
{
    unsigned long long wg = REMAP ? xcd_remap(blockIdx.x, gridDim.x) : blockIdx.x;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int rep = 0; rep < 2; ++rep) {
        v4f acc = {0, 0, 0, 0};
        for (int s = 0; s < WAVES; ++s) {
            const unsigned long long sym = (wg * 2 + rep) * WAVES + s;
            if (sym >= n_sym) break;
            const v4f* src = in + (sym * IN_STRIDE_ROWS + 2) * 64;
            for (int i = w; i < IN_ROWS; i += WAVES) acc += src[i * 64 + lane];
        }
        for (int s = 0; s < WAVES; ++s) {
            const unsigned long long sym = (wg * 2 + rep) * WAVES + s;
            if (sym >= n_sym) break;
            v4f* dst = out + sym * OUT_ROWS * 64;
            for (int i = w; i < OUT_ROWS; i += WAVES) __builtin_nontemporal_store(acc, dst + i * 64 + lane);
        }
    }
}
// S3: row-granular grid: wave g moves rows g, g + G, ... (reads half as many rows as it writes)
__global__ void __launch_bounds__(256) s3(const v4f* __restrict__ in, v4f* __restrict__ out, unsigned long long n_sym)
{
    const unsigned long long g = (unsigned long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const unsigned long long out_rows = n_sym * OUT_ROWS;
    if (g * 2 >= out_rows) return;
    const unsigned long long r = g;                       // input row index among payload rows
    const unsigned long long sym = r / IN_ROWS, ir = r % IN_ROWS;
    const v4f v = in[(sym * IN_STRIDE_ROWS + 2 + ir) * 64 + lane];
    __builtin_nontemporal_store(v, out + (2 * g) * 64 + lane);
    __builtin_nontemporal_store(v, out + (2 * g + 1) * 64 + lane);
}
// read-only and write-only row-granular streams, for reference
__global__ void __launch_bounds__(256) rd_only(const v4f* __restrict__ in, v4f* __restrict__ out, unsigned long long rows)
{
    const unsigned long long g = (unsigned long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    v4f acc = {0, 0, 0, 0};
    for (int i = 0; i < 8; ++i) { const unsigned long long r = g * 8 + i; if (r < rows) acc += in[r * 64 + lane]; }
    if (acc.x == 12345.678f) out[lane] = acc;           // never true: keeps the loads alive
}
__global__ void __launch_bounds__(256) wr_only(v4f* __restrict__ out, unsigned long long rows)
{
    const unsigned long long g = (unsigned long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const v4f v = {1.f, 2.f, 3.f, (float)g};
    for (int i = 0; i < 8; ++i) { const unsigned long long r = g * 8 + i; if (r < rows) __builtin_nontemporal_store(v, out + r * 64 + lane); }
}
template <typename F> float timeit(F&& f)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    f(); f();
    CK(hipEventRecord(e0));
    for (int i = 0; i < 6; ++i) f();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms / 6;
}
int main()
{
    const unsigned long long n_sym = 1ull << 19;
    const size_t in_bytes = n_sym * IN_STRIDE_ROWS * 1024, out_bytes = n_sym * OUT_ROWS * 1024;
    char *in, *out; CK(hipMalloc(&in, in_bytes)); CK(hipMalloc(&out, out_bytes)); CK(hipMemset(in, 0, in_bytes));
    const double bytes = (double)n_sym * (IN_ROWS + OUT_ROWS) * 1024;
    auto report = [&](const char* name, float ms) { printf("%-44s %7.3f ms %7.1f GB/s\n", name, ms, bytes / ms / 1e6); fflush(stdout); };
    for (int rep = 0; rep < 2; ++rep) {
#define RUN0(W, R) report("S0 wave/symbol, " #W " waves/WG, remap " #R, timeit([&] { s0<W, R><<<(unsigned)((n_sym + 2 * W - 1) / (2 * W)), 64 * W>>>((const v4f*)in, (v4f*)out, n_sym); }))
#define RUN1(W, R) report("S1 shared symbols, " #W " waves/WG, remap " #R, timeit([&] { s1<W, R><<<(unsigned)((n_sym + 2 * W - 1) / (2 * W)), 64 * W>>>((const v4f*)in, (v4f*)out, n_sym); }))
        RUN0(1, 1); RUN0(1, 0); RUN0(4, 1); RUN0(4, 0);
        RUN1(4, 1); RUN1(4, 0); RUN1(8, 1); RUN1(8, 0); RUN1(16, 0);
        {
            const unsigned long long rows = out_bytes / 1024;
            const unsigned grid = (unsigned)((rows / 8 + 3) / 4);
            float ms = timeit([&] { rd_only<<<grid, 256>>>((const v4f*)out, (v4f*)in, rows); });
            printf("%-44s %7.3f ms %7.1f GB/s\n", "read only, 8 rows per wave", ms, (double)out_bytes / ms / 1e6);
            ms = timeit([&] { wr_only<<<grid, 256>>>((v4f*)out, rows); });
            printf("%-44s %7.3f ms %7.1f GB/s\n", "write only (NT), 8 rows per wave", ms, (double)out_bytes / ms / 1e6);
        }
        report("S3 row-granular grid", timeit([&] { s3<<<(unsigned)((n_sym * IN_ROWS + 3) / 4), 256>>>((const v4f*)in, (v4f*)out, n_sym); }));
    }
    return 0;
}
