// membench7.hip -- what a software-pipelined wave could reach: the 64-QAM receive kernel's traffic
// (8 KiB read : 24 KiB written per symbol, one wave per symbol, whole-grid XCD-aware order) where each
// wave handles a RUN of consecutive symbols and requests symbol k+1 before it works on symbol k, with
// ALU work standing in for the transform (ALU dependent v4f FMAs per symbol), at 2..16 resident waves
// per CU (dynamic LDS as the throttle).  RUN = 1 is the shipped kernel's shape (no prefetch).
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
template <bool PREFETCH>
__global__ void __launch_bounds__(64) pipe_copy(const v4f* __restrict__ in, v4f* __restrict__ out, unsigned long long n_sym, int run, int alu)
{
    extern __shared__ char pad[];
    const unsigned long long wg = xcd_remap(blockIdx.x, gridDim.x);
    const int lane = threadIdx.x;
    const unsigned long long s0 = wg * (unsigned long long)run;
    if (s0 >= n_sym) return;
    const unsigned long long s1 = s0 + run < n_sym ? s0 + run : n_sym;
    v4f nxt[IN_ROWS];
    {
        const v4f* src = in + (s0 * IN_STRIDE_ROWS + 2) * 64;
#pragma unroll
        for (int i = 0; i < IN_ROWS; ++i) nxt[i] = src[i * 64 + lane];
    }
    for (unsigned long long sym = s0; sym < s1; ++sym) {
        v4f cur[IN_ROWS];
#pragma unroll
        for (int i = 0; i < IN_ROWS; ++i) cur[i] = nxt[i];
        if (PREFETCH && sym + 1 < s1) {
            const v4f* src = in + ((sym + 1) * IN_STRIDE_ROWS + 2) * 64;
#pragma unroll
            for (int i = 0; i < IN_ROWS; ++i) nxt[i] = src[i * 64 + lane];
        }
        v4f a = cur[0] + cur[1], b = cur[2] + cur[3], c = cur[4] + cur[5], d = cur[6] + cur[7];
        for (int it = 0; it < alu; ++it) {                  // 4 independent v4f FMA chains = 16 VALU lane-ops (8 packed) per iteration
            a = a * 1.0001f + b; b = b * 0.9999f + c; c = c * 1.0002f + d; d = d * 0.9998f + a;
        }
        const v4f r = a + b + c + d;
        v4f* dst = out + sym * OUT_ROWS * 64;
#pragma unroll
        for (int i = 0; i < OUT_ROWS; ++i) __builtin_nontemporal_store(r, dst + i * 64 + lane);
        if (!PREFETCH && sym + 1 < s1) {
            const v4f* src = in + ((sym + 1) * IN_STRIDE_ROWS + 2) * 64;
#pragma unroll
            for (int i = 0; i < IN_ROWS; ++i) nxt[i] = src[i * 64 + lane];
        }
    }
}
template <typename F> float timeit(F&& f)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    f(); f();
    CK(hipEventRecord(e0));
    for (int i = 0; i < 5; ++i) f();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms / 5;
}
int main(int argc, char** argv)
{
    const unsigned long long n_sym = 1ull << 19;
    const size_t in_bytes = n_sym * IN_STRIDE_ROWS * 1024, out_bytes = n_sym * OUT_ROWS * 1024;
    char *in, *out; CK(hipMalloc(&in, in_bytes)); CK(hipMalloc(&out, out_bytes)); CK(hipMemset(in, 0, in_bytes));
    const double bytes = (double)n_sym * (IN_ROWS + OUT_ROWS) * 1024;
    const int lds_for[] = {0, 20480, 26624, 32768, 40960, 53248, 81920};      // 16, 8, 6, 5, 4, 3, 2 workgroups per CU
    const int wpc[] = {16, 8, 6, 5, 4, 3, 2};
    const int alus[] = {0, 110, 220};                                          // x 8 packed (16 plain) VALU instructions
    const int runs[] = {1, 4, 16, 64};
    for (int ai = 0; ai < 3; ++ai)
        for (int ri = 0; ri < 4; ++ri)
            for (int pf = 0; pf < 2; ++pf) {
                if (runs[ri] == 1 && pf == 1) continue;
                printf("alu %4d run %3d prefetch %d:", alus[ai] * 8, runs[ri], pf);
                for (int li = 0; li < 7; ++li) {
                    const int run = runs[ri], alu = alus[ai], lds = lds_for[li];
                    const unsigned grid = (unsigned)((n_sym + run - 1) / run);
                    float ms = pf ? timeit([&] { pipe_copy<true><<<grid, 64, lds>>>((const v4f*)in, (v4f*)out, n_sym, run, alu); })
                                  : timeit([&] { pipe_copy<false><<<grid, 64, lds>>>((const v4f*)in, (v4f*)out, n_sym, run, alu); });
                    printf("  %2d w/CU %6.0f", wpc[li], bytes / ms / 1e6);
                }
                printf("\n"); fflush(stdout);
            }
    return 0;
}
