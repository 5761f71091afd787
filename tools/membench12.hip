// membench12.hip -- write rate per allocation when many 24 GiB allocations are alive at once: is the
// fast state (7.1 instead of 6.5 TB/s) tied to where in HBM an allocation lands?
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
int main(int argc, char** argv)
{
    const size_t GB = 1ull << 30;
    const size_t gib = argc > 1 ? strtoull(argv[1], 0, 10) : 24;
    const int k = argc > 2 ? atoi(argv[2]) : 9;
    const size_t bytes = gib * GB;
    size_t fr, tot; CK(hipMemGetInfo(&fr, &tot));
    printf("free %.1f GiB of %.1f GiB\n", fr / (double)GB, tot / (double)GB);
    for (int round = 0; round < 3; ++round) {
        char* b[16];
        for (int i = 0; i < k; ++i) { CK(hipMalloc(&b[i], bytes)); CK(hipMemset(b[i], 0, bytes)); }
        CK(hipDeviceSynchronize());
        printf("round %d:", round);
        for (int i = 0; i < k; ++i) {
            const unsigned grid = (unsigned)(bytes / 4096);
            float ms = timeit([&] { wr_streams<<<grid, 64>>>((v4f*)b[i], bytes / 8 / 1024); }, 3);
            printf(" %5.0f", bytes / ms / 1e6);
        }
        printf("  GB/s   (va:");
        for (int i = 0; i < k; ++i) printf(" %llx", (unsigned long long)b[i] >> 30);
        printf(")\n"); fflush(stdout);
        // free in a different order every round
        for (int i = 0; i < k; ++i) CK(hipFree(b[(i * (round + 2)) % k == 0 && i ? i : (i * 5 + round) % k == i ? i : i]));
    }
    return 0;
}
