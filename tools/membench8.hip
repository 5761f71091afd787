// membench8.hip -- why do launches of the same copy differ by 10 % between allocations?  For a few
// fresh allocations: the 64-QAM receive-kernel traffic shape (one wave per symbol, whole-grid XCD-aware
// order), timed, plus one instrumented launch that records when each XCD finished its eighth
// (s_memrealtime, 100 MHz; XCC id from the hardware register).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float v4f __attribute__((ext_vector_type(4)));
constexpr int IN_ROWS = 8, OUT_ROWS = 24, IN_STRIDE_ROWS = 10;

__device__ __forceinline__ unsigned long long xcd_remap(unsigned long long wg, unsigned long long grid, unsigned swz)
{
    if (swz >= 8) return wg;                       // natural order
    const unsigned long long win = grid & ~7ull;
    if (wg < win) wg = ((wg & 7) ^ swz) * (win >> 3) + (wg >> 3);
    return wg;
}
template <bool STAMP>
__global__ void __launch_bounds__(64) sym_copy(const v4f* __restrict__ in, v4f* __restrict__ out, unsigned long long n_sym, unsigned long long* stamps, unsigned swz)
{
    extern __shared__ char pad[];
    const unsigned long long sym = xcd_remap(blockIdx.x, gridDim.x, swz);
    if (sym >= n_sym) return;
    const int lane = threadIdx.x;
    unsigned long long t0 = 0;
    if (STAMP) t0 = __builtin_amdgcn_s_memrealtime();
    const v4f* src = in + (sym * IN_STRIDE_ROWS + 2) * 64;
    v4f acc = {0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < IN_ROWS; ++i) acc += src[i * 64 + lane];
    v4f* dst = out + sym * OUT_ROWS * 64;
#pragma unroll
    for (int i = 0; i < OUT_ROWS; ++i) __builtin_nontemporal_store(acc, dst + i * 64 + lane);
    if (STAMP) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        if (lane == 0) { stamps[2 * blockIdx.x] = (t0 << 3) | (xcc & 7u); stamps[2 * blockIdx.x + 1] = t1; }
    }
}
template <typename F> float timeit(F&& f, int n)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    f();
    CK(hipEventRecord(e0));
    for (int i = 0; i < n; ++i) f();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms / n;
}
int main(int argc, char** argv)
{
    const unsigned long long n_sym = argc > 1 ? strtoull(argv[1], 0, 10) : (1ull << 20);
    const int lds = argc > 2 ? atoi(argv[2]) : 0;
    const unsigned swz = argc > 3 ? atoi(argv[3]) : 0;
    const size_t in_bytes = n_sym * IN_STRIDE_ROWS * 1024, out_bytes = n_sym * OUT_ROWS * 1024;
    const double bytes = (double)n_sym * (IN_ROWS + OUT_ROWS) * 1024;
    unsigned long long* stamps; CK(hipMalloc(&stamps, n_sym * 16));
    std::vector<unsigned long long> h(n_sym * 2);
    for (int trial = 0; trial < 8; ++trial) {
        char *in, *out; CK(hipMalloc(&in, in_bytes)); CK(hipMalloc(&out, out_bytes)); CK(hipMemset(in, 0, in_bytes)); CK(hipMemset(out, 0, out_bytes));
        float ms = timeit([&] { sym_copy<false><<<(unsigned)n_sym, 64, lds>>>((const v4f*)in, (v4f*)out, n_sym, nullptr, swz); }, 4);
        float ms2 = timeit([&] { sym_copy<true><<<(unsigned)n_sym, 64, lds>>>((const v4f*)in, (v4f*)out, n_sym, stamps, swz); }, 2);
        CK(hipMemcpy(h.data(), stamps, n_sym * 16, hipMemcpyDeviceToHost));
        unsigned long long t0 = ~0ull, fin[8] = {0}, cnt[8] = {0}, rfin[8] = {0};
        unsigned rx[8] = {0};
        for (unsigned long long b = 0; b < n_sym; ++b) {
            const unsigned x = h[2 * b] & 7u; const unsigned long long s0 = h[2 * b] >> 3, e1 = h[2 * b + 1];
            if (s0 < t0) t0 = s0;
            if (e1 > fin[x]) fin[x] = e1;
            if (e1 > rfin[b & 7]) rfin[b & 7] = e1;
            rx[b & 7] |= 1u << x;
            cnt[x]++;
        }
        printf("trial %d in %p out %p: %7.1f GB/s (stamped %7.1f) | XCD finish us:", trial, (void*)in, (void*)out, bytes / ms / 1e6, bytes / ms2 / 1e6);
        for (int x = 0; x < 8; ++x) printf(" %5.0f", (fin[x] - t0) / 100.0);
        if (trial < 2) {
            // progress per XCD: workgroups finished per 0.5 ms bin
            const int NB = 16; unsigned long long hist[8][NB] = {};
            for (unsigned long long b = 0; b < n_sym; ++b) { const unsigned x = h[2 * b] & 7u; const double us = (h[2 * b + 1] - t0) / 100.0; int bin = (int)(us / 500.0); if (bin >= NB) bin = NB - 1; hist[x][bin]++; }
            for (int x = 0; x < 8; ++x) { printf("\n   XCD %d finished per 0.5 ms:", x); for (int i = 0; i < NB; ++i) printf(" %6llu", hist[x][i]); }
            printf("\n");
        }
        printf(" | by region (id%%8):");
        for (int x = 0; x < 8; ++x) printf(" %5.0f", (rfin[x] - t0) / 100.0);
        printf(" | XCDs seen per region:");
        for (int x = 0; x < 8; ++x) printf(" %02x", rx[x]);
        printf("\n"); fflush(stdout);
        CK(hipFree(in)); CK(hipFree(out));
    }
    return 0;
}
