// membench14.hip -- eight per-XCD write streams: does a small per-stream SKEW (stream i starts i * skew
// bytes later, so the streams are never at the same low address bits at the same time) reach the fast
// state that large spans reach by luck?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float v4f __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(64) wr_streams(v4f* __restrict__ out, unsigned long long stride_rows, unsigned long long skew_rows)
{
    const unsigned long long x = blockIdx.x & 7;
    const unsigned long long r0 = x * (stride_rows + skew_rows) + (unsigned long long)(blockIdx.x >> 3) * 4;
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
    const size_t GB = 1ull << 30, KB = 1024;
    const size_t skews[] = {0, 1 * KB, 4 * KB, 16 * KB, 64 * KB, 256 * KB, 1024 * KB, 1028 * KB, 4096 * KB, 16384 * KB, 65536 * KB, 65540 * KB, 262144 * KB};
    for (size_t gib : {8, 24}) {
        for (int trial = 0; trial < 3; ++trial) {
            char* buf; CK(hipMalloc(&buf, (gib + 3) * GB)); CK(hipMemset(buf, 0, (gib + 3) * GB));
            printf("%2zu GiB trial %d:", gib, trial);
            for (size_t sk : skews) {
                const size_t bytes = gib * GB;
                const unsigned grid = (unsigned)(bytes / 4096);
                float ms = timeit([&] { wr_streams<<<grid, 64>>>((v4f*)buf, bytes / 8 / 1024, sk / 1024); }, 3);
                printf(" %5.0f", bytes / ms / 1e6);
            }
            printf("\n"); fflush(stdout);
            CK(hipFree(buf));
        }
    }
    printf("skews (KiB):"); for (size_t sk : skews) printf(" %zu", sk / KB); printf("\n");
    return 0;
}
