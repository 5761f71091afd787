// membench5.hip -- write-only streams: which store policy / shape reaches the hipMemset fill rate?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float v4f __attribute__((ext_vector_type(4)));

template <int POLICY>
__device__ __forceinline__ void store16(v4f* p, v4f v)
{
    if constexpr (POLICY == 0) *p = v;
    else if constexpr (POLICY == 1) __builtin_nontemporal_store(v, p);
    else if constexpr (POLICY == 2) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(p), "v"(v) : "memory");
    else if constexpr (POLICY == 3) asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");
    else if constexpr (POLICY == 4) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" :: "v"(p), "v"(v) : "memory");
    else if constexpr (POLICY == 5) asm volatile("global_store_dwordx4 %0, %1, off sc0" :: "v"(p), "v"(v) : "memory");
}
// ROWS consecutive 1 KiB rows per wave (ROWS = 16: the receive kernel's region per symbol at 16-QAM)
template <int POLICY, int ROWS, int REMAP = 0>
__global__ void __launch_bounds__(64) wr_rows(v4f* __restrict__ out, unsigned long long rows)
{
    unsigned long long g = blockIdx.x;
    if (REMAP) {
        const unsigned long long win = 4096, full = ((unsigned long long)gridDim.x / win) * win;
        if (g < full) { const unsigned long long r = g % win; g = g - r + (r & 7) * (win >> 3) + (r >> 3); }
    }
    const int lane = threadIdx.x;
    const v4f v = {1.f, 2.f, 3.f, (float)g};
#pragma unroll
    for (int i = 0; i < ROWS; ++i) { const unsigned long long r = g * ROWS + i; if (r < rows) store16<POLICY>(out + r * 64 + lane, v); }
}
// grid-stride: wave g writes rows g, g + G, g + 2G, ...
template <int POLICY>
__global__ void __launch_bounds__(256) wr_stride(v4f* __restrict__ out, unsigned long long rows)
{
    const unsigned long long G = (unsigned long long)gridDim.x * 4;
    const int lane = threadIdx.x & 63;
    const v4f v = {1.f, 2.f, 3.f, 4.f};
    for (unsigned long long r = (unsigned long long)blockIdx.x * 4 + (threadIdx.x >> 6); r < rows; r += G) store16<POLICY>(out + r * 64 + lane, v);
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
    const size_t bytes = (size_t)8 << 30;
    char* out; CK(hipMalloc(&out, bytes));
    const unsigned long long rows = bytes / 1024;
    auto rep = [&](const char* name, float ms) { printf("%-46s %7.3f ms %7.1f GB/s\n", name, ms, bytes / ms / 1e6); fflush(stdout); };
    for (int r = 0; r < 2; ++r) {
        rep("hipMemsetAsync", timeit([&] { CK(hipMemsetAsync(out, 0, bytes, 0)); }));
#define ROWSRUN(P, R) rep("wave writes " #R " rows, policy " #P, timeit([&] { wr_rows<P, R><<<(unsigned)((rows + R - 1) / R), 64>>>((v4f*)out, rows); }))
        ROWSRUN(0, 16); ROWSRUN(1, 16); ROWSRUN(2, 16); ROWSRUN(3, 16); ROWSRUN(4, 16); ROWSRUN(5, 16);
        ROWSRUN(0, 1); ROWSRUN(1, 1); ROWSRUN(1, 4);
#define ROWSRUNX(P, R) rep("wave writes " #R " rows, XCD order, policy " #P, timeit([&] { wr_rows<P, R, 1><<<(unsigned)((rows + R - 1) / R), 64>>>((v4f*)out, rows); }))
        ROWSRUN(0, 2); ROWSRUN(0, 4); ROWSRUN(0, 8); ROWSRUN(1, 8); ROWSRUN(0, 24); ROWSRUN(1, 24); ROWSRUN(0, 32);
        ROWSRUNX(0, 2); ROWSRUNX(0, 4); ROWSRUNX(1, 4); ROWSRUNX(0, 8); ROWSRUNX(1, 8); ROWSRUNX(0, 16); ROWSRUNX(1, 16); ROWSRUNX(0, 24); ROWSRUNX(1, 24);
#define STRIDERUN(P, B) rep("grid-stride rows, " #B " blocks, policy " #P, timeit([&] { wr_stride<P><<<B, 256>>>((v4f*)out, rows); }))
        STRIDERUN(0, 2048); STRIDERUN(1, 2048); STRIDERUN(2, 2048); STRIDERUN(0, 8192); STRIDERUN(1, 8192);
    }
    return 0;
}
