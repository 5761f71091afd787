// membench16.hip -- does a THIRD, small read stream cost the receive kernel's traffic shape more than its bytes?
// Shape: 64-QAM at N = 1024, one wave per symbol (8 KiB read, 24 KiB written, XCD-aware order).  Variants:
//   0  no extra stream
//   1  every wave also reads its set's 12 KiB "taps" (one set per 14 symbols) AHEAD of its samples and folds them in
//   2  the same reads issued AFTER the samples
//   3  every wave reads the taps of set 0 (always cache-resident): the request path without the HBM bytes
//   4  as 1, but only the FIRST wave of each set reads them (HBM bytes of the taps without most of the requests)
// Rate is counted on the 32 KiB of samples + LLRs only, like bench.py's roofline.frac.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float v4f __attribute__((ext_vector_type(4)));
constexpr int IN_ROWS = 8, OUT_ROWS = 24, IN_STRIDE_ROWS = 10, TAP_ROWS = 12, PER = 14;
__device__ __forceinline__ unsigned long long xcd_remap(unsigned long long wg, unsigned long long grid)
{
    const unsigned long long win = grid & ~7ull;
    if (wg < win) wg = (wg & 7) * (win >> 3) + (wg >> 3);
    return wg;
}
template <int MODE>
__global__ void __launch_bounds__(64) sym_copy(const v4f* __restrict__ in, const v4f* __restrict__ taps, v4f* __restrict__ out, unsigned long long n_sym)
{
    extern __shared__ char pad[];
    const unsigned long long sym = xcd_remap(blockIdx.x, gridDim.x);
    if (sym >= n_sym) return;
    const int lane = threadIdx.x;
    const v4f* src = in + (sym * IN_STRIDE_ROWS + 2) * 64;
    const unsigned long long set = MODE == 3 ? 0 : sym / PER;
    const v4f* tp = taps + set * TAP_ROWS * 64;
    const bool want = MODE == 1 || MODE == 2 || MODE == 3 || (MODE == 4 && sym % PER == 0);
    v4f acc = {0, 0, 0, 0};
    v4f t[TAP_ROWS];
    if (MODE != 2 && MODE != 0) {
        if (want) {
#pragma unroll
            for (int i = 0; i < TAP_ROWS; ++i) t[i] = tp[i * 64 + lane];
        }
    }
    v4f x[IN_ROWS];
#pragma unroll
    for (int i = 0; i < IN_ROWS; ++i) x[i] = src[i * 64 + lane];
    if (MODE == 2) {
#pragma unroll
        for (int i = 0; i < TAP_ROWS; ++i) t[i] = tp[i * 64 + lane];
    }
#pragma unroll
    for (int i = 0; i < IN_ROWS; ++i) acc += x[i];
    v4f* dst = out + sym * OUT_ROWS * 64;
    if (MODE != 0 && want) {
#pragma unroll
        for (int i = 0; i < OUT_ROWS; ++i) __builtin_nontemporal_store(acc * t[i % TAP_ROWS], dst + i * 64 + lane);
    } else {
#pragma unroll
        for (int i = 0; i < OUT_ROWS; ++i) __builtin_nontemporal_store(acc, dst + i * 64 + lane);
    }
}
template <typename F> float timeit(F&& f, int n)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    f(); f();
    CK(hipEventRecord(e0));
    for (int i = 0; i < n; ++i) f();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1)); return ms / n;
}
int main(int argc, char** argv)
{
    const unsigned long long n_sym = argc > 1 ? strtoull(argv[1], 0, 10) : (1ull << 20);
    const size_t in_bytes = n_sym * IN_STRIDE_ROWS * 1024, out_bytes = n_sym * OUT_ROWS * 1024, tap_bytes = (n_sym / PER + 1) * TAP_ROWS * 1024;
    char *in, *out, *taps;
    CK(hipMalloc(&in, in_bytes)); CK(hipMalloc(&taps, tap_bytes)); CK(hipMalloc(&out, out_bytes));
    CK(hipMemset(in, 0, in_bytes)); CK(hipMemset(taps, 0, tap_bytes)); CK(hipMemset(out, 0, out_bytes));
    const double bytes = (double)n_sym * (IN_ROWS + OUT_ROWS) * 1024;
    const int pads[] = {0, 16384, 24576};      // dynamic LDS: 16-32 / 8 / 5 waves per CU
    for (int rep = 0; rep < 2; ++rep)
        for (int lds : pads) {
            float ms[5];
            ms[0] = timeit([&] { sym_copy<0><<<(unsigned)n_sym, 64, lds>>>((const v4f*)in, (const v4f*)taps, (v4f*)out, n_sym); }, 5);
            ms[1] = timeit([&] { sym_copy<1><<<(unsigned)n_sym, 64, lds>>>((const v4f*)in, (const v4f*)taps, (v4f*)out, n_sym); }, 5);
            ms[2] = timeit([&] { sym_copy<2><<<(unsigned)n_sym, 64, lds>>>((const v4f*)in, (const v4f*)taps, (v4f*)out, n_sym); }, 5);
            ms[3] = timeit([&] { sym_copy<3><<<(unsigned)n_sym, 64, lds>>>((const v4f*)in, (const v4f*)taps, (v4f*)out, n_sym); }, 5);
            ms[4] = timeit([&] { sym_copy<4><<<(unsigned)n_sym, 64, lds>>>((const v4f*)in, (const v4f*)taps, (v4f*)out, n_sym); }, 5);
            printf("lds %5d:  none %6.1f   taps ahead %6.1f   taps after %6.1f   set 0 only %6.1f   first wave of the set %6.1f   GB/s of samples + LLRs\n", lds,
                   bytes / ms[0] / 1e6, bytes / ms[1] / 1e6, bytes / ms[2] / 1e6, bytes / ms[3] / 1e6, bytes / ms[4] / 1e6);
            fflush(stdout);
        }
    return 0;
}
