// membench10.hip -- is the good / bad state a property of one of the allocations?  Per fresh allocation
// pair: read-only sweep of each buffer, write-only fill of each buffer (XCD order, 4 rows per wave),
// and the mixed 8:24 KiB per-symbol copy in both directions of buffer use.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned long long xcd_remap(unsigned long long wg, unsigned long long grid)
{
    const unsigned long long win = grid & ~7ull;
    if (wg < win) wg = (wg & 7) * (win >> 3) + (wg >> 3);
    return wg;
}
template <int ROWS> __global__ void __launch_bounds__(64) rd_rows(const v4f* __restrict__ in, v4f* sink, unsigned long long rows)
{
    const unsigned long long g = xcd_remap(blockIdx.x, gridDim.x);
    v4f acc = {0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < ROWS; ++i) { const unsigned long long r = g * ROWS + i; if (r < rows) acc += in[r * 64 + threadIdx.x]; }
    if (acc.x == 12345.678f) sink[threadIdx.x] = acc;
}
template <int ROWS> __global__ void __launch_bounds__(64) wr_rows(v4f* __restrict__ out, unsigned long long rows)
{
    const unsigned long long g = xcd_remap(blockIdx.x, gridDim.x);
    const v4f v = {1.f, 2.f, 3.f, (float)g};
#pragma unroll
    for (int i = 0; i < ROWS; ++i) { const unsigned long long r = g * ROWS + i; if (r < rows) __builtin_nontemporal_store(v, out + r * 64 + threadIdx.x); }
}
template <int IN_ROWS, int OUT_ROWS> __global__ void __launch_bounds__(64) mixed(const v4f* __restrict__ in, v4f* __restrict__ out, unsigned long long n)
{
    const unsigned long long s = xcd_remap(blockIdx.x, gridDim.x);
    if (s >= n) return;
    v4f acc = {0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < IN_ROWS; ++i) acc += in[(s * IN_ROWS + i) * 64 + threadIdx.x];
#pragma unroll
    for (int i = 0; i < OUT_ROWS; ++i) __builtin_nontemporal_store(acc, out + (s * OUT_ROWS + i) * 64 + threadIdx.x);
}
template <typename F> float timeit(F&& f, int n)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    f();
    CK(hipEventRecord(e0));
    for (int i = 0; i < n; ++i) f();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1)); return ms / n;
}
int main(int argc, char** argv)
{
    const unsigned long long n = 1ull << 20;
    const size_t a_bytes = n * 8 * 1024, b_bytes = n * 24 * 1024;       // 8 GiB and 24 GiB
    for (int trial = 0; trial < 8; ++trial) {
        char *a, *b; CK(hipMalloc(&a, a_bytes)); CK(hipMalloc(&b, b_bytes)); CK(hipMemset(a, 0, a_bytes)); CK(hipMemset(b, 0, b_bytes));
        const unsigned long long ra = a_bytes / 1024, rb = b_bytes / 1024;
        float t_ra = timeit([&] { rd_rows<4><<<(unsigned)(ra / 4), 64>>>((const v4f*)a, (v4f*)b, ra); }, 3);
        float t_rb = timeit([&] { rd_rows<4><<<(unsigned)(rb / 4), 64>>>((const v4f*)b, (v4f*)a, rb); }, 3);
        float t_wa = timeit([&] { wr_rows<4><<<(unsigned)(ra / 4), 64>>>((v4f*)a, ra); }, 3);
        float t_wb = timeit([&] { wr_rows<4><<<(unsigned)(rb / 4), 64>>>((v4f*)b, rb); }, 3);
        float t_ab = timeit([&] { mixed<8, 24><<<(unsigned)n, 64>>>((const v4f*)a, (v4f*)b, n); }, 3);
        float t_ba = timeit([&] { mixed<24, 8><<<(unsigned)n, 64>>>((const v4f*)b, (v4f*)a, n); }, 3);
        printf("trial %d a %p b %p | read a %6.0f b %6.0f | write a %6.0f b %6.0f | a->b (8:24) %6.0f | b->a (24:8) %6.0f GB/s\n", trial, (void*)a, (void*)b,
               a_bytes / t_ra / 1e6, b_bytes / t_rb / 1e6, a_bytes / t_wa / 1e6, b_bytes / t_wb / 1e6, (a_bytes + b_bytes) / t_ab / 1e6, (a_bytes + b_bytes) / t_ba / 1e6);
        fflush(stdout);
        CK(hipFree(a)); CK(hipFree(b));
    }
    return 0;
}
