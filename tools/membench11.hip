// membench11.hip -- eight per-XCD write streams inside ONE allocation: does the spacing between the
// streams (the size of the eighth each XCD works through) set the write rate?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float v4f __attribute__((ext_vector_type(4)));
// wave b: XCD stream (b & 7), position (b >> 3) in it; 4 rows of 1 KiB per wave
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
int main()
{
    const size_t MB = 1ull << 20, GB = 1ull << 30;
    const size_t total = 28 * GB;
    char* buf; CK(hipMalloc(&buf, total)); CK(hipMemset(buf, 0, total));
    const size_t len = 2 * GB;                                   // bytes each stream writes
    const unsigned grid = (unsigned)(8 * (len / 4096));
    printf("buf %p\n", (void*)buf);
    auto run = [&](size_t stride) {
        float ms = timeit([&] { wr_streams<<<grid, 64>>>((v4f*)buf, stride / 1024); }, 3);
        printf("stride %8.3f MiB (%6.3f GiB): %7.1f GB/s\n", stride / (double)MB, stride / (double)GB, 8.0 * len / ms / 1e6); fflush(stdout);
    };
    for (size_t s = 2048; s <= 3328; s += 64) run(s * MB);
    const size_t odd[] = {3072 * MB + 4096, 3072 * MB + 64 * 1024, 3072 * MB + 1 * MB, 3072 * MB + 2 * MB, 3072 * MB + 6 * MB, 3072 * MB + 10 * MB, 3072 * MB + 18 * MB, 3072 * MB + 34 * MB,
                          2304 * MB + 2 * MB, 2304 * MB + 18 * MB, 2560 * MB + 34 * MB, 3000 * MB, 3100 * MB, 2900 * MB + 512 * 1024};
    for (size_t s : odd) run(s);
    return 0;
}
