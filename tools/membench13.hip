// membench13.hip -- is the fast-write state a property of physical regions?  One large pool; the write
// rate (eight per-XCD streams) of every 8 GiB slice of it; then separate allocations of several sizes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float v4f __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(64) wr_streams(v4f* __restrict__ out, unsigned long long stride_rows)
{
    const unsigned long long r0 = (unsigned long long)(blockIdx.x & 7) * stride_rows + (unsigned long long)(blockIdx.x >> 3) * 4;
    const v4f v = {1.f, 2.f, 3.f, (float)blockIdx.x};
#pragma unroll
    for (int i = 0; i < 4; ++i) __builtin_nontemporal_store(v, out + (r0 + i) * 64 + threadIdx.x);
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
static double rate(char* p, size_t bytes)
{
    const unsigned grid = (unsigned)(bytes / 4096);
    float ms = timeit([&] { wr_streams<<<grid, 64>>>((v4f*)p, bytes / 8 / 1024); }, 3);
    return bytes / ms / 1e6;
}
int main()
{
    const size_t GB = 1ull << 30;
    {
        const size_t pool_gib = 192, slice = 8;
        char* pool; CK(hipMalloc(&pool, pool_gib * GB)); CK(hipMemset(pool, 0, pool_gib * GB));
        printf("pool %p, %zu GiB; write rate of each %zu GiB slice:\n", (void*)pool, pool_gib, slice);
        for (size_t s = 0; s < pool_gib / slice; ++s) printf(" %5.0f", rate(pool + s * slice * GB, slice * GB));
        printf("\n24 GiB windows at 8 GiB steps:\n");
        for (size_t s = 0; s + 3 <= pool_gib / slice; ++s) printf(" %5.0f", rate(pool + s * slice * GB, 24 * GB));
        printf("\nwhole pool: %5.0f\n", rate(pool, pool_gib * GB)); fflush(stdout);
        CK(hipFree(pool));
    }
    const size_t sizes[] = {8, 12, 16, 20, 24, 32, 48};
    for (size_t gib : sizes) {
        const int k = (int)(216 / gib) > 12 ? 12 : (int)(216 / gib);
        char* b[16];
        for (int i = 0; i < k; ++i) { CK(hipMalloc(&b[i], gib * GB)); CK(hipMemset(b[i], 0, gib * GB)); }
        printf("%2zu GiB x %2d:", gib, k);
        for (int i = 0; i < k; ++i) printf(" %5.0f", rate(b[i], gib * GB));
        printf("\n"); fflush(stdout);
        for (int i = 0; i < k; ++i) CK(hipFree(b[i]));
    }
    return 0;
}
