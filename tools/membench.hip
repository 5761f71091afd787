// membench.hip -- memory-system ceilings for the access patterns of the fused receive kernel.
// No arithmetic: each wave reads the 8 KiB payload of a 10 KiB OFDM symbol (CP skipped) and writes
// OUT_F4 float4 per lane per 128-bin block in one of two store shapes:
//   shape 0: "per-lane chunk"  -- lane t writes CH consecutive float4 at (block*64 + t)*CH   (what the kernel does)
//   shape 1: "fully coalesced" -- store i writes float4 at block*64*CH + i*64 + t
// Build: hipcc -O3 --offload-arch=gfx950 tools/membench.hip -o tools/membench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int CH, int SHAPE, int WAVES, bool PREFETCH>
__global__ void __launch_bounds__(64 * WAVES) stream_kernel(const float4* __restrict__ in, float4* __restrict__ out, unsigned long long n_sym)
{
    const int lane = threadIdx.x & 63;
    const unsigned long long wave = (unsigned long long)blockIdx.x * WAVES + (threadIdx.x >> 6);
    const unsigned long long n_waves = (unsigned long long)gridDim.x * WAVES;
    float4 v[8], vn[8];
    auto load = [&](unsigned long long s, float4 (&d)[8]) {
        const float4* src = in + s * 640 + 128 + lane;   // 10 KiB per symbol, 2 KiB CP skipped
#pragma unroll
        for (int q = 0; q < 8; ++q) d[q] = src[q * 64];
    };
    if (PREFETCH && wave < n_sym) load(wave, vn);
    for (unsigned long long s = wave; s < n_sym; s += n_waves) {
        if (PREFETCH) {
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = vn[q];
            if (s + n_waves < n_sym) load(s + n_waves, vn);
        } else {
            load(s, v);
        }
        float4* dst = out + s * (512 * CH);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                float4 w = v[q];
                w.x += (float)c;
                if (SHAPE == 0) dst[(q * 64 + lane) * CH + c] = w;
                else dst[q * 64 * CH + c * 64 + lane] = w;
            }
        }
    }
}

// non-persistent form: one wave per symbol, grid = n_sym / WAVES; NT = nontemporal stores
template <int CH, int SHAPE, int WAVES, bool NT>
__global__ void __launch_bounds__(64 * WAVES) oneshot_kernel(const float4* __restrict__ in, float4* __restrict__ out, unsigned long long n_sym)
{
    const int lane = threadIdx.x & 63;
    const unsigned long long s = (unsigned long long)blockIdx.x * WAVES + (threadIdx.x >> 6);
    if (s >= n_sym) return;
    float4 v[8];
    const float4* src = in + s * 640 + 128 + lane;
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = src[q * 64];
    float4* dst = out + s * (512 * CH);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            float4 w = v[q];
            w.x += (float)c;
            float4* a = (SHAPE == 0) ? &dst[(q * 64 + lane) * CH + c] : &dst[q * 64 * CH + c * 64 + lane];
            typedef float v4f __attribute__((ext_vector_type(4)));
            if (NT) { v4f t4 = {w.x, w.y, w.z, w.w}; __builtin_nontemporal_store(t4, reinterpret_cast<v4f*>(a)); } else *a = w;
        }
    }
}

// two symbols per wave, NT loads + NT stores, coalesced
template <int CH, int WAVES, bool NTL, int SPWV>
__global__ void __launch_bounds__(64 * WAVES) oneshot2_kernel(const float4* __restrict__ in, float4* __restrict__ out, unsigned long long n_sym)
{
    typedef float v4f __attribute__((ext_vector_type(4)));
    const int lane = threadIdx.x & 63;
    const unsigned long long s0 = ((unsigned long long)blockIdx.x * WAVES + (threadIdx.x >> 6)) * SPWV;
    v4f v[SPWV][8];
#pragma unroll
    for (int u = 0; u < SPWV; ++u) {
        if (s0 + u >= n_sym) return;
        const v4f* src = reinterpret_cast<const v4f*>(in) + (s0 + u) * 640 + 128 + lane;
#pragma unroll
        for (int q = 0; q < 8; ++q) v[u][q] = NTL ? __builtin_nontemporal_load(src + q * 64) : src[q * 64];
    }
#pragma unroll
    for (int u = 0; u < SPWV; ++u) {
        v4f* dst = reinterpret_cast<v4f*>(out) + (s0 + u) * (512 * CH);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                v4f w = v[u][q];
                w.x += (float)c;
                __builtin_nontemporal_store(w, dst + q * 64 * CH + c * 64 + lane);
            }
        }
    }
}

template <int CH, int WAVES, bool NTL, int SPWV>
void run2(const float4* in, float4* out, unsigned long long n_sym, const char* name)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const unsigned grid = (unsigned)((n_sym + WAVES * SPWV - 1) / (WAVES * SPWV));
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((oneshot2_kernel<CH, WAVES, NTL, SPWV>), dim3(grid), dim3(64 * WAVES), 0, 0, in, out, n_sym);
    CK(hipEventRecord(e0));
    const int reps = 5;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((oneshot2_kernel<CH, WAVES, NTL, SPWV>), dim3(grid), dim3(64 * WAVES), 0, 0, in, out, n_sym);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    const double bytes = (double)n_sym * (8192.0 + 8192.0 * CH);
    printf("%-52s %7.3f ms  %7.1f GB/s (%4.1f %% of 8 TB/s)\n", name, ms, bytes / ms / 1e6, bytes / ms / 1e6 / 80.0);
}

template <int CH, int SHAPE, int WAVES, bool NT>
void run1(const float4* in, float4* out, unsigned long long n_sym, const char* name)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const unsigned grid = (unsigned)((n_sym + WAVES - 1) / WAVES);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((oneshot_kernel<CH, SHAPE, WAVES, NT>), dim3(grid), dim3(64 * WAVES), 0, 0, in, out, n_sym);
    CK(hipEventRecord(e0));
    const int reps = 5;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((oneshot_kernel<CH, SHAPE, WAVES, NT>), dim3(grid), dim3(64 * WAVES), 0, 0, in, out, n_sym);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    const double bytes = (double)n_sym * (8192.0 + 8192.0 * CH);
    printf("%-52s %7.3f ms  %7.1f GB/s (%4.1f %% of 8 TB/s)\n", name, ms, bytes / ms / 1e6, bytes / ms / 1e6 / 80.0);
}

__global__ void flat_copy(const float4* __restrict__ in, float4* __restrict__ out, size_t n)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = in[i];
}

template <int CH, int SHAPE, int WAVES, bool PREFETCH>
void run(const float4* in, float4* out, unsigned long long n_sym, int wg_per_cu, const char* name)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int grid = 256 * wg_per_cu;
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((stream_kernel<CH, SHAPE, WAVES, PREFETCH>), dim3(grid), dim3(64 * WAVES), 0, 0, in, out, n_sym);
    CK(hipEventRecord(e0));
    const int reps = 5;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((stream_kernel<CH, SHAPE, WAVES, PREFETCH>), dim3(grid), dim3(64 * WAVES), 0, 0, in, out, n_sym);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    const double bytes = (double)n_sym * (8192.0 + 8192.0 * CH);
    printf("%-44s wg/cu %2d  %7.3f ms  %7.1f GB/s (%4.1f %% of 8 TB/s)\n", name, wg_per_cu, ms, bytes / ms / 1e6, bytes / ms / 1e6 / 80.0);
}

int main()
{
    const unsigned long long n_sym = 200000;
    float4 *in, *out;
    CK(hipMalloc(&in, n_sym * 10240));
    CK(hipMalloc(&out, n_sym * 8192 * 4));
    CK(hipMemset(in, 0, n_sym * 10240));
    {
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        const size_t n = n_sym * 512;
        for (int g : {2048, 8192, 65536}) {
            flat_copy<<<g, 256>>>(in, out, n);
            CK(hipEventRecord(e0));
            for (int i = 0; i < 5; ++i) flat_copy<<<g, 256>>>(in, out, n);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
            printf("flat float4 copy grid %6d: %7.1f GB/s\n", g, 2.0 * n * 16 / ms / 1e6);
        }
    }
    run1<1, 1, 1, false>(in, out, n_sym, "oneshot 1w: wr 8K coalesced");
    run1<1, 1, 4, false>(in, out, n_sym, "oneshot 4w: wr 8K coalesced");
    run1<1, 1, 4, true>(in, out, n_sym, "oneshot 4w: wr 8K coalesced NT");
    run1<3, 0, 1, false>(in, out, n_sym, "oneshot 1w: wr 24K per-lane chunks");
    run1<3, 1, 1, false>(in, out, n_sym, "oneshot 1w: wr 24K coalesced");
    run1<3, 0, 4, false>(in, out, n_sym, "oneshot 4w: wr 24K per-lane chunks");
    run1<3, 1, 4, false>(in, out, n_sym, "oneshot 4w: wr 24K coalesced");
    run1<3, 0, 4, true>(in, out, n_sym, "oneshot 4w: wr 24K per-lane chunks NT");
    run1<3, 1, 4, true>(in, out, n_sym, "oneshot 4w: wr 24K coalesced NT");
    run1<2, 0, 4, false>(in, out, n_sym, "oneshot 4w: wr 16K per-lane chunks");
    run1<2, 1, 4, false>(in, out, n_sym, "oneshot 4w: wr 16K coalesced");
    run1<2, 1, 4, true>(in, out, n_sym, "oneshot 4w: wr 16K coalesced NT");
    run2<3, 1, false, 1>(in, out, n_sym, "v2 1w: 24K NT st");
    run2<3, 1, true, 1>(in, out, n_sym, "v2 1w: 24K NT st + NT ld");
    run2<3, 4, true, 1>(in, out, n_sym, "v2 4w: 24K NT st + NT ld");
    run2<3, 1, true, 2>(in, out, n_sym, "v2 1w x2sym: 24K NT st + NT ld");
    run2<3, 4, false, 2>(in, out, n_sym, "v2 4w x2sym: 24K NT st");
    run2<2, 1, true, 1>(in, out, n_sym, "v2 1w: 16K NT st + NT ld");
    run2<1, 1, true, 1>(in, out, n_sym, "v2 1w: 8K NT st + NT ld");
    run2<4, 1, true, 1>(in, out, n_sym, "v2 1w: 32K NT st + NT ld");
    run2<4, 1, false, 1>(in, out, n_sym, "v2 1w: 32K NT st");
    return 0;
}
