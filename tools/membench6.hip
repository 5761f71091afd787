// membench6.hip -- the 64-QAM receive kernel's traffic (8 KiB read : 24 KiB written per symbol, one wave
// per symbol, whole-grid XCD-aware order) under different load / store cache policies and numbers of
// resident waves, plus a plain 1:1 float4 copy for calibration against the guide's 6.29 TB/s.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float v4f __attribute__((ext_vector_type(4)));

constexpr int IN_ROWS = 8, OUT_ROWS = 24, IN_STRIDE_ROWS = 10;

template <int POLICY>
__device__ __forceinline__ void store16(v4f* p, v4f v)
{
    if constexpr (POLICY == 0) *p = v;
    else if constexpr (POLICY == 1) __builtin_nontemporal_store(v, p);
    else if constexpr (POLICY == 2) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(p), "v"(v) : "memory");
    else if constexpr (POLICY == 3) asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");
    else if constexpr (POLICY == 4) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" :: "v"(p), "v"(v) : "memory");
}
template <int POLICY>
__device__ __forceinline__ v4f load16(const v4f* p)
{
    if constexpr (POLICY == 0) return *p;
    else if constexpr (POLICY == 1) return __builtin_nontemporal_load(p);
    else { v4f v; asm volatile("global_load_dwordx4 %0, %1, off sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory"); return v; }
}
__device__ __forceinline__ unsigned long long xcd_remap(unsigned long long wg, unsigned long long grid)
{
    const unsigned long long win = grid & ~7ull;
    if (wg < win) { wg = (wg & 7) * (win >> 3) + (wg >> 3); }
    return wg;
}
template <int LP, int SP, int REMAP>
__global__ void __launch_bounds__(64) sym_copy(const v4f* __restrict__ in, v4f* __restrict__ out, unsigned long long n_sym)
{
    extern __shared__ char pad[];
    const unsigned long long sym = REMAP ? xcd_remap(blockIdx.x, gridDim.x) : blockIdx.x;
    if (sym >= n_sym) return;
    const int lane = threadIdx.x;
    const v4f* src = in + (sym * IN_STRIDE_ROWS + 2) * 64;
    v4f acc = {0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < IN_ROWS; ++i) acc += load16<LP>(src + i * 64 + lane);
    v4f* dst = out + sym * OUT_ROWS * 64;
#pragma unroll
    for (int i = 0; i < OUT_ROWS; ++i) store16<SP>(dst + i * 64 + lane, acc);
}
__global__ void __launch_bounds__(256) copy11(const v4f* __restrict__ in, v4f* __restrict__ out, unsigned long long n)
{
    for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * 256) out[i] = in[i];
}
__global__ void __launch_bounds__(256) copy11_flat(const v4f* __restrict__ in, v4f* __restrict__ out, unsigned long long n)
{
    const unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) __builtin_nontemporal_store(in[i], out + i);
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
    auto report = [&](const char* name, int lds, float ms) { printf("%-40s lds %6d %7.3f ms %7.1f GB/s\n", name, lds, ms, bytes / ms / 1e6); fflush(stdout); };
    for (int rep = 0; rep < 2; ++rep) {
        {
            const unsigned long long n = in_bytes / 16;
            float ms = timeit([&] { copy11<<<256 * 16, 256>>>((const v4f*)in, (v4f*)out, n); });
            printf("%-40s %7.3f ms %7.1f GB/s\n", "1:1 float4 copy, grid-stride 4096 WG", ms, 2.0 * in_bytes / ms / 1e6);
            ms = timeit([&] { copy11_flat<<<(unsigned)((n + 255) / 256), 256>>>((const v4f*)in, (v4f*)out, n); });
            printf("%-40s %7.3f ms %7.1f GB/s\n", "1:1 float4 copy, one element per thread NT", ms, 2.0 * in_bytes / ms / 1e6);
        }
#define RUN(LP, SP, R, LDS) report("sym copy load " #LP " store " #SP " remap " #R, LDS, timeit([&] { sym_copy<LP, SP, R><<<(unsigned)n_sym, 64, LDS>>>((const v4f*)in, (v4f*)out, n_sym); }))
        RUN(0, 1, 1, 0); RUN(0, 0, 1, 0); RUN(0, 2, 1, 0); RUN(0, 3, 1, 0); RUN(0, 4, 1, 0);
        RUN(1, 1, 1, 0); RUN(1, 3, 1, 0); RUN(2, 1, 1, 0); RUN(2, 3, 1, 0);
        RUN(0, 1, 0, 0);
        RUN(0, 1, 1, 10240); RUN(0, 1, 1, 20480); RUN(0, 1, 1, 40960); RUN(0, 3, 1, 20480); RUN(0, 3, 1, 40960);
    }
    return 0;
}
