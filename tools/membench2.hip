// membench2.hip -- second round of memory-system experiments for the fused receive kernel's
// traffic shape (read 8 KiB of every 10 KiB, write 24 KiB per OFDM symbol, one wave per symbol).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float v4f __attribute__((ext_vector_type(4)));

// MODE 0: read+write, 1: write only, 2: read only.  REMAP: 0 none, 1: symbol = bit-mix of block id
template <int CH, int MODE, int WAVES, int REMAP, int SYMS>
__global__ void __launch_bounds__(64 * WAVES) k(const v4f* __restrict__ in, v4f* __restrict__ out, unsigned long long n_sym, float* sink)
{
    const int lane = threadIdx.x & 63;
    unsigned long long w = (unsigned long long)blockIdx.x * WAVES + (threadIdx.x >> 6);
    if (REMAP == 1) {   // consecutive workgroups 8 apart in symbol space within windows of 4096
        const unsigned long long win = w >> 12, r = w & 4095;
        w = (win << 12) | ((r & 7) << 9) | (r >> 3);
    }
    float acc = 0.f;
    for (int u = 0; u < SYMS; ++u) {
        const unsigned long long s = w * SYMS + u;
        if (s >= n_sym) break;
        v4f v[8];
        if (MODE != 1) {
            const v4f* src = in + s * 640 + 128 + lane;
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = src[q * 64];
        } else {
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = v4f{(float)q, (float)lane, 1.f, 2.f};
        }
        if (MODE != 2) {
            v4f* dst = out + s * (512 * CH);
#pragma unroll
            for (int q = 0; q < 8; ++q)
#pragma unroll
                for (int c = 0; c < CH; ++c) { v4f t = v[q]; t.x += (float)c; __builtin_nontemporal_store(t, dst + q * 64 * CH + c * 64 + lane); }
        } else {
#pragma unroll
            for (int q = 0; q < 8; ++q) acc += v[q].x + v[q].w;
        }
    }
    if (MODE == 2 && acc == 12345.678f) sink[0] = acc;
}

template <int CH, int MODE, int WAVES, int REMAP, int SYMS>
void run(const v4f* in, v4f* out, unsigned long long n_sym, float* sink, const char* name)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const unsigned grid = (unsigned)((n_sym + WAVES * SYMS - 1) / (WAVES * SYMS));
    for (int i = 0; i < 2; ++i) k<CH, MODE, WAVES, REMAP, SYMS><<<grid, 64 * WAVES>>>(in, out, n_sym, sink);
    CK(hipEventRecord(e0));
    for (int i = 0; i < 6; ++i) k<CH, MODE, WAVES, REMAP, SYMS><<<grid, 64 * WAVES>>>(in, out, n_sym, sink);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 6;
    const double bytes = (double)n_sym * ((MODE != 1 ? 8192.0 : 0.0) + (MODE != 2 ? 8192.0 * CH : 0.0));
    printf("%-50s %7.3f ms %7.1f GB/s\n", name, ms, bytes / ms / 1e6);
}

int main()
{
    const unsigned long long n_sym = 262144;
    char *in, *out; float* sink;
    CK(hipMalloc(&in, n_sym * 10240 + (8 << 20)));
    CK(hipMalloc(&out, n_sym * 8192 * 3 + (8 << 20)));
    CK(hipMalloc(&sink, 64));
    CK(hipMemset(in, 1, n_sym * 10240));
    for (int rep = 0; rep < 2; ++rep) {
        run<3, 0, 1, 0, 1>((v4f*)in, (v4f*)out, n_sym, sink, "R8K W24K");
        run<3, 1, 1, 0, 1>((v4f*)in, (v4f*)out, n_sym, sink, "W24K only");
        run<3, 2, 1, 0, 1>((v4f*)in, (v4f*)out, n_sym, sink, "R8K/10K only");
        run<3, 0, 1, 1, 1>((v4f*)in, (v4f*)out, n_sym, sink, "R8K W24K remap (XCD-contiguous)");
        run<3, 0, 4, 0, 1>((v4f*)in, (v4f*)out, n_sym, sink, "R8K W24K 4 waves/WG");
        run<3, 0, 1, 0, 2>((v4f*)in, (v4f*)out, n_sym, sink, "R8K W24K 2 symbols per wave");
        run<3, 0, 1, 0, 1>((v4f*)in, (v4f*)(out + 4096), n_sym, sink, "R8K W24K out +4 KiB");
        run<3, 0, 1, 0, 1>((v4f*)in, (v4f*)(out + 65536 + 256), n_sym, sink, "R8K W24K out +64 KiB+256");
        run<3, 0, 1, 0, 1>((v4f*)(in + 2048), (v4f*)(out + (1 << 20) + 2048), n_sym, sink, "R8K W24K in +2 KiB out +1 MiB+2 KiB");
        run<2, 0, 1, 0, 1>((v4f*)in, (v4f*)out, n_sym, sink, "R8K W16K");
        run<1, 0, 1, 0, 1>((v4f*)in, (v4f*)out, n_sym, sink, "R8K W8K");
    }
    return 0;
}
