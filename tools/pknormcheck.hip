// What v_cvt_pknorm_u16_f32 computes, bit for bit: every float in [0, 1] and a few specials against candidate
// CPU-expressible formulas.  hipcc --offload-arch=gfx950 -O2 tools/pknormcheck.hip -o /tmp/pknormcheck
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
__device__ unsigned hw(float x) { u16x2 q = __builtin_amdgcn_cvt_pknorm_u16(x, 0.0f); return q.x; }
__global__ void sweep(unsigned long long* bad)
{
    const unsigned long long total = 0x3F800000ull + 1;
    unsigned long long a = 0, b = 0, c = 0, d = 0;
    for (unsigned long long i = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; i < total; i += (unsigned long long)gridDim.x * blockDim.x) {
        const float x = __uint_as_float((unsigned)i);
        const unsigned h = hw(x);
        a += h != (unsigned)rint((double)x * 65535.0);
        b += h != (unsigned)rintf(x * 65535.0f);
        c += h != (unsigned)(x * 65535.0f + 0.5f);
        d += h != (unsigned)((double)x * 65535.0 + 0.5);
    }
    atomicAdd(&bad[0], a); atomicAdd(&bad[1], b); atomicAdd(&bad[2], c); atomicAdd(&bad[3], d);
}
__global__ void specials(const float* in, unsigned* out, int n) { int i = threadIdx.x; if (i < n) out[i] = hw(in[i]); }
int main()
{
    unsigned long long* bad; hipMalloc(&bad, 32); hipMemset(bad, 0, 32);
    sweep<<<4096, 256>>>(bad);
    unsigned long long h[4]; hipMemcpy(h, bad, 32, hipMemcpyDeviceToHost);
    printf("mismatches over all floats in [0,1]: RNE(exact product) %llu, RNE(float product) %llu, half-up(float product) %llu, half-up(exact product) %llu\n", h[0], h[1], h[2], h[3]);
    float sp[12] = {-0.0f, -1.0f, 1.0f, 1.0000001f, 2.0f, INFINITY, -INFINITY, NAN, 1e-40f, 0.5f / 65535.0f, 1.5f / 65535.0f, 2.5f / 65535.0f};
    float* din; unsigned* dout; hipMalloc(&din, 48); hipMalloc(&dout, 48); hipMemcpy(din, sp, 48, hipMemcpyHostToDevice);
    specials<<<1, 64>>>(din, dout, 12);
    unsigned o[12]; hipMemcpy(o, dout, 48, hipMemcpyDeviceToHost);
    for (int i = 0; i < 12; ++i) printf("pknorm(%g) = %u\n", sp[i], o[i]);
    return 0;
}
