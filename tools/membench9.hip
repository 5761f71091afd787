// membench9.hip -- one pool, the read buffer at its start and the write buffer `delta` bytes behind the
// read buffer's end: does the RELATIVE placement of the two streams (beyond 2 MiB) set the rate?
// Shape: the 64-QAM receive kernel's traffic, one wave per symbol, whole-grid XCD-aware order.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float v4f __attribute__((ext_vector_type(4)));
constexpr int IN_ROWS = 8, OUT_ROWS = 24, IN_STRIDE_ROWS = 10;
__device__ __forceinline__ unsigned long long xcd_remap(unsigned long long wg, unsigned long long grid)
{
    const unsigned long long win = grid & ~7ull;
    if (wg < win) wg = (wg & 7) * (win >> 3) + (wg >> 3);
    return wg;
}
__global__ void __launch_bounds__(64) sym_copy(const v4f* __restrict__ in, v4f* __restrict__ out, unsigned long long n_sym)
{
    extern __shared__ char pad[];
    const unsigned long long sym = xcd_remap(blockIdx.x, gridDim.x);
    if (sym >= n_sym) return;
    const int lane = threadIdx.x;
    const v4f* src = in + (sym * IN_STRIDE_ROWS + 2) * 64;
    v4f acc = {0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < IN_ROWS; ++i) acc += src[i * 64 + lane];
    v4f* dst = out + sym * OUT_ROWS * 64;
#pragma unroll
    for (int i = 0; i < OUT_ROWS; ++i) __builtin_nontemporal_store(acc, dst + i * 64 + lane);
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
    const unsigned long long n_sym = argc > 1 ? strtoull(argv[1], 0, 10) : (1ull << 20);
    const int lds = argc > 2 ? atoi(argv[2]) : 0;
    const size_t MB = 1ull << 20;
    const size_t in_bytes = n_sym * IN_STRIDE_ROWS * 1024, out_bytes = n_sym * OUT_ROWS * 1024;
    const size_t in_span = ((in_bytes + 2 * MB - 1) / (2 * MB)) * 2 * MB;
    const size_t max_delta = 6144 * MB;
    char* pool; CK(hipMalloc(&pool, in_span + out_bytes + max_delta + 2 * MB));
    CK(hipMemset(pool, 0, in_span + out_bytes + max_delta));
    const double bytes = (double)n_sym * (IN_ROWS + OUT_ROWS) * 1024;
    printf("pool %p\n", (void*)pool);
    auto run = [&](size_t delta) {
        char* in = pool; char* out = pool + in_span + delta;
        float ms = timeit([&] { sym_copy<<<(unsigned)n_sym, 64, lds>>>((const v4f*)in, (v4f*)out, n_sym); }, 3);
        printf("delta %6zu MiB: %7.1f GB/s\n", delta / MB, bytes / ms / 1e6); fflush(stdout);
    };
    for (size_t d = 0; d <= 64; d += 2) run(d * MB);
    for (size_t d = 96; d <= 1024; d += 32) run(d * MB);
    for (size_t d = 1280; d <= 6144; d += 256) run(d * MB);
    return 0;
}
