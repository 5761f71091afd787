// valubench.hip -- issue cost of the VALU instructions the ALU-bound kernels (frame generator,
// Monte-Carlo link, hard-only receive) are made of, in SIMD cycles per wave64 instruction at full
// occupancy: the weights behind the "VALU-issue" roofline of DESIGN.md section 6.
// Each mode runs 8 independent dependency chains of one instruction, 4096 x 8 instructions per wave.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float v2f __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void __launch_bounds__(64) valu(unsigned* out, long long* clk, int rounds, unsigned seed)
{
    const long long c0 = clock64(), t0 = wall_clock64();      // shader clock / constant 100 MHz clock
    unsigned a[8];
    unsigned long long w[8];
    float f[8];
    v2f p[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        a[i] = threadIdx.x * 2654435761u + i + seed;
        w[i] = a[i];
        f[i] = 1.0f + 1e-3f * static_cast<float>((threadIdx.x + i) & 15);
        p[i] = v2f{f[i], f[i] + 0.5f};
    }
    const unsigned m = 0xD2511F53u + seed;
    const float g = 1.0000001f + static_cast<float>(seed);
#pragma unroll 16
    for (int r = 0; r < rounds; ++r) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if constexpr (MODE == 10) a[i] = a[i] + (a[i] >> 3);                                        // v_lshrrev + v_add (or one v_lshl_add-class op): full-rate reference
            else if constexpr (MODE == 0) a[i] = a[i] * m;                                                   // v_mul_lo_u32
            else if constexpr (MODE == 1) a[i] = __umulhi(a[i], m);                                     // v_mul_hi_u32
            else if constexpr (MODE == 2) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(w[i]) : "v"(a[i]), "v"(m) : "vcc");   // v_mad_u64_u32, nothing else
            else if constexpr (MODE == 11) asm volatile("v_cmp_lt_u32 vcc, %1, %2\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(m), "v"(a[(i + 1) & 7]) : "vcc");   // compare + select
            else if constexpr (MODE == 12) asm volatile("v_bfe_u32 %0, %0, 3, 29" : "+v"(a[i]));              // VOP3 integer op
            else if constexpr (MODE == 13) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(f[i]) : "v"(g));          // unpacked fp32 multiply
            else if constexpr (MODE == 3) a[i] = (a[i] ^ m) + 0x9E3779B9u;                              // v_xor + v_add (2 instructions)
            else if constexpr (MODE == 4) f[i] = f[i] * g;                                              // v_mul_f32
            else if constexpr (MODE == 5) p[i] = p[i] * v2f{g, g};                                      // v_pk_mul_f32
            else if constexpr (MODE == 6) f[i] = __builtin_fmaf(f[i], g, 1e-7f);                        // v_fma_f32
            else if constexpr (MODE == 7) f[i] = __builtin_amdgcn_sqrtf(f[i] + 1.0f);                   // v_sqrt_f32 (+ v_add)
            else if constexpr (MODE == 8) f[i] = __builtin_rintf(f[i] * g);                             // v_rndne (+ v_mul)
            else if constexpr (MODE == 9) a[i] = (a[i] & 0xFFFFFFu) * (m & 0xFFFFFFu);                  // 24-bit multiply candidate
        }
    }
    unsigned acc = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) acc ^= a[i] ^ static_cast<unsigned>(w[i]) ^ static_cast<unsigned>(w[i] >> 32) ^ __float_as_uint(f[i]) ^ __float_as_uint(p[i].x) ^ __float_as_uint(p[i].y);
    out[blockIdx.x * 64 + threadIdx.x] = acc;
    if (blockIdx.x == 4096 && threadIdx.x == 0) { clk[0] = clock64() - c0; clk[1] = wall_clock64() - t0; }
}

template <int MODE> void run(const char* name, int per_iter, unsigned* out, int blocks, int rounds)
{
    static long long* clk = nullptr;
    if (!clk) CK(hipMalloc(&clk, 16));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(valu<MODE>, dim3(blocks), dim3(64), 0, 0, out, clk, rounds, 0u);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(valu<MODE>, dim3(blocks), dim3(64), 0, 0, out, clk, rounds, 0u);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    long long h[2]; CK(hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost));
    const double ghz = h[1] > 0 ? double(h[0]) / double(h[1]) * 0.1 : 0.0;      // shader clock while this kernel ran
    const double cyc = ms * 1e-3 * ghz * 1e9 / (double(blocks) / 1024.0 * rounds * 8.0);
    printf("%-34s %8.3f ms  shader clock %5.3f GHz  %6.2f SIMD cycles per chain step (%d instruction%s)\n", name, ms, ghz, cyc, per_iter, per_iter > 1 ? "s" : "");
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
}

int main(int argc, char** argv)
{
    const int blocks = 1024 * 16, rounds = argc > 1 ? atoi(argv[1]) : 4096;
    unsigned* out; CK(hipMalloc(&out, size_t(blocks) * 64 * 4));
    run<10>("reference: shift + add", 2, out, blocks, rounds);
    run<0>("v_mul_lo_u32", 1, out, blocks, rounds);
    run<1>("v_mul_hi_u32", 1, out, blocks, rounds);
    run<2>("v_mad_u64_u32", 1, out, blocks, rounds);
    run<11>("v_cmp_lt_u32 + v_cndmask_b32", 2, out, blocks, rounds);
    run<12>("v_bfe_u32 (VOP3)", 1, out, blocks, rounds);
    run<13>("v_mul_f32 (unpacked)", 1, out, blocks, rounds);
    run<3>("v_xor_b32 + v_add_u32", 2, out, blocks, rounds);
    run<4>("v_mul_f32", 1, out, blocks, rounds);
    run<5>("v_pk_mul_f32", 1, out, blocks, rounds);
    run<6>("v_fma_f32", 1, out, blocks, rounds);
    run<7>("v_add_f32 + v_sqrt_f32", 2, out, blocks, rounds);
    run<8>("v_mul_f32 + v_rndne_f32", 2, out, blocks, rounds);
    run<9>("v_and + v_mul_u32_u24", 2, out, blocks, rounds);
    CK(hipFree(out));
    return 0;
}
