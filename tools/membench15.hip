// membench15.hip -- write rate against the NUMBER of concurrent streams (1, 2, 4, 8: XCDs that share a
// stream interleave their 4 KiB pieces in it) and the span of the buffer, several allocations each.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float v4f __attribute__((ext_vector_type(4)));
// ns streams; XCD x = blockIdx & 7 belongs to stream x / (8 / ns); piece index inside the stream = q * (8 / ns) + x % (8 / ns)
__global__ void __launch_bounds__(64) wr_streams(v4f* __restrict__ out, unsigned long long stride_rows, unsigned ns)
{
    const unsigned x = blockIdx.x & 7, share = 8 / ns;
    const unsigned long long q = blockIdx.x >> 3;
    const unsigned long long r0 = (unsigned long long)(x / share) * stride_rows + (q * share + x % share) * 4;
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
int main()
{
    const size_t GB = 1ull << 30;
    for (size_t gib : {8, 24, 48}) {
        for (int trial = 0; trial < 4; ++trial) {
            char* buf; CK(hipMalloc(&buf, gib * GB)); CK(hipMemset(buf, 0, gib * GB));
            printf("%2zu GiB trial %d: streams 1/2/4/8:", gib, trial);
            for (unsigned ns : {1u, 2u, 4u, 8u}) {
                const size_t bytes = gib * GB;
                const unsigned grid = (unsigned)(bytes / 4096);
                float ms = timeit([&] { wr_streams<<<grid, 64>>>((v4f*)buf, bytes / ns / 1024, ns); }, 3);
                printf(" %5.0f", bytes / ms / 1e6);
            }
            printf("\n"); fflush(stdout);
            CK(hipFree(buf));
        }
    }
    return 0;
}
