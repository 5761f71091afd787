// membench3.hip -- write-path experiments: what store shape reaches the fill rate (~6.9 TB/s)?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float v4f __attribute__((ext_vector_type(4)));

// each wave writes KB KiB contiguous; NT: nontemporal; REMAP as membench2; ORDER 0: ascending 1 KiB rows
template <int KB, int NT, int WAVES, int REMAP>
__global__ void __launch_bounds__(64 * WAVES) wr(v4f* __restrict__ out, unsigned long long n_chunks)
{
    const int lane = threadIdx.x & 63;
    unsigned long long w = (unsigned long long)blockIdx.x * WAVES + (threadIdx.x >> 6);
    if (REMAP == 1) { const unsigned long long win = w >> 12, r = w & 4095; w = (win << 12) | ((r & 7) << 9) | (r >> 3); }
    if (w >= n_chunks) return;
    v4f* dst = out + w * (KB * 64);
    v4f t = {(float)lane, 1.f, 2.f, (float)w};
#pragma unroll
    for (int i = 0; i < KB; ++i) {
        if (NT) __builtin_nontemporal_store(t, dst + i * 64 + lane); else dst[i * 64 + lane] = t;
    }
}
// torch-like: 256 threads, each thread 4 x float4 at stride 4 KiB, block covers 16 KiB
template <int NT>
__global__ void __launch_bounds__(256) wr_torch(v4f* __restrict__ out, unsigned long long n_vec)
{
    const unsigned long long base = (unsigned long long)blockIdx.x * 1024;
    v4f t = {1.f, 2.f, 3.f, 4.f};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const unsigned long long i = base + j * 256 + threadIdx.x;
        if (i < n_vec) { if (NT) __builtin_nontemporal_store(t, out + i); else out[i] = t; }
    }
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
template <int KB, int NT, int WAVES, int REMAP>
void run(v4f* out, size_t bytes, const char* name)
{
    const unsigned long long n = bytes / (KB * 1024);
    const unsigned grid = (unsigned)((n + WAVES - 1) / WAVES);
    float ms = timeit([&] { wr<KB, NT, WAVES, REMAP><<<grid, 64 * WAVES>>>(out, n); });
    printf("%-50s %7.3f ms %7.1f GB/s\n", name, ms, (double)n * KB * 1024 / ms / 1e6);
}
int main()
{
    const size_t bytes = (size_t)6 << 30;
    char* out; CK(hipMalloc(&out, bytes));
    for (int rep = 0; rep < 2; ++rep) {
        { float ms = timeit([&] { CK(hipMemsetAsync(out, 0, bytes, 0)); }); printf("%-50s %7.3f ms %7.1f GB/s\n", "hipMemsetAsync", ms, bytes / ms / 1e6); }
        { const unsigned long long nv = bytes / 16; float ms = timeit([&] { wr_torch<0><<<(unsigned)(nv / 1024), 256>>>((v4f*)out, nv); }); printf("%-50s %7.3f ms %7.1f GB/s\n", "torch-like 256thr x4 strided", ms, bytes / ms / 1e6); }
        { const unsigned long long nv = bytes / 16; float ms = timeit([&] { wr_torch<1><<<(unsigned)(nv / 1024), 256>>>((v4f*)out, nv); }); printf("%-50s %7.3f ms %7.1f GB/s\n", "torch-like NT", ms, bytes / ms / 1e6); }
        run<24, 1, 1, 0>((v4f*)out, bytes, "wave 24K NT");
        run<24, 0, 1, 0>((v4f*)out, bytes, "wave 24K plain");
        run<24, 1, 1, 1>((v4f*)out, bytes, "wave 24K NT remap");
        run<24, 0, 1, 1>((v4f*)out, bytes, "wave 24K plain remap");
        run<8, 0, 1, 0>((v4f*)out, bytes, "wave 8K plain");
        run<4, 0, 1, 0>((v4f*)out, bytes, "wave 4K plain");
        run<4, 0, 4, 0>((v4f*)out, bytes, "4 waves x 4K plain (16K/WG)");
        run<4, 1, 4, 0>((v4f*)out, bytes, "4 waves x 4K NT (16K/WG)");
        run<24, 0, 4, 0>((v4f*)out, bytes, "4 waves x 24K plain");
        run<24, 1, 4, 1>((v4f*)out, bytes, "4 waves x 24K NT remap");
        run<1, 0, 4, 0>((v4f*)out, bytes, "4 waves x 1K plain (4K/WG)");
    }
    return 0;
}
