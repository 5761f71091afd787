// xchgbench.hip -- what does it cost a wave to move FFT data between lanes?  A/B of the exchange
// primitives gfx950 offers, on the receive kernel's shape at N = 1024: one wave holds a symbol as
// 16 complex values = 32 dwords per lane, and each exchange between register passes swaps FOUR
// register-index bits with four lane-index bits (a 16 x 16 transpose inside the wave).
//   lds     the shipped way: 8 x ds_write2_b64 + 8 x ds_read2_b64 per lane (padded rows), any permutation at once
//   swap32  v_permlane32_swap: one register bit <-> lane bit 5, 16 instructions per bit
//   swap16  v_permlane16_swap: one register bit <-> lane bit 4, 16 instructions per bit
//   dpp8    v_mov_dpp row_ror:8 with bank masks: one register bit <-> lane bit 3
//   dpp1    v_mov_dpp quad_perm [1,0,3,2] + v_cndmask: one register bit <-> lane bit 0
//   bperm   ds_bpermute_b32: 32 instructions per arbitrary permutation
// Every mode runs `rounds` exchanges per wave at full occupancy (no memory traffic) and reports the
// time per exchange step and wave, in SIMD cycles at 2.4 GHz (4 SIMDs per CU, 256 CUs).
// A full 4-bit transpose costs 1 x lds, or (say) swap32 + swap16 + dpp8 + dpp-low.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int R = 32;      // dwords per lane

template <int MODE>
__global__ void __launch_bounds__(64) xchg(unsigned* out, int rounds)
{
    __shared__ __attribute__((aligned(16))) unsigned lds[64 * 17 * 2];
    const unsigned lane = threadIdx.x;
    unsigned v[R];
#pragma unroll
    for (int i = 0; i < R; ++i) v[i] = lane * 131u + i * 7u + blockIdx.x;
    for (int r = 0; r < rounds; ++r) {
        if constexpr (MODE == 0) {
            // rows of 16 values padded to 17 (conflict-free both ways), read back transposed inside each group of 16 lanes
            uint2* w = reinterpret_cast<uint2*>(lds);
#pragma unroll
            for (int j = 0; j < R / 2; ++j) w[lane * 17 + j] = make_uint2(v[2 * j], v[2 * j + 1]);
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int i = 0; i < R / 2; ++i) {
                // value i of lane l <- value (l & 15) of lane ((l & 48) | i): register bits <-> low lane bits
                const uint2 q = w[((lane & 48u) | static_cast<unsigned>(i)) * 17 + (lane & 15u)];
                v[2 * i] = q.x; v[2 * i + 1] = q.y;
            }
            __builtin_amdgcn_wave_barrier();
        } else if constexpr (MODE == 1) {
#pragma unroll
            for (int i = 0; i < R / 2; ++i) {
                auto q = __builtin_amdgcn_permlane32_swap(v[i], v[i + R / 2], false, false);
                v[i] = q[0]; v[i + R / 2] = q[1];
            }
        } else if constexpr (MODE == 2) {
#pragma unroll
            for (int i = 0; i < R / 2; ++i) {
                auto q = __builtin_amdgcn_permlane16_swap(v[i], v[i + R / 2], false, false);
                v[i] = q[0]; v[i + R / 2] = q[1];
            }
        } else if constexpr (MODE == 3) {
#pragma unroll
            for (int i = 0; i < R / 2; ++i) {
                const unsigned a = v[i], b = v[i + R / 2];
                // lanes with bit 3 clear (banks 0,1) take the partner's a into b; lanes with bit 3 set take the partner's b into a
                v[i + R / 2] = __builtin_amdgcn_update_dpp(b, a, 0x128 /* row_ror:8 */, 0xF, 0x3, false);
                v[i] = __builtin_amdgcn_update_dpp(a, b, 0x128, 0xF, 0xC, false);
            }
        } else if constexpr (MODE == 4) {
#pragma unroll
            for (int i = 0; i < R / 2; ++i) {
                const unsigned a = v[i], b = v[i + R / 2];
                const unsigned pa = __builtin_amdgcn_update_dpp(0u, a, 0xB1 /* quad_perm [1,0,3,2] */, 0xF, 0xF, false);
                const unsigned pb = __builtin_amdgcn_update_dpp(0u, b, 0xB1, 0xF, 0xF, false);
                v[i] = (lane & 1u) ? pb : a;
                v[i + R / 2] = (lane & 1u) ? b : pa;
            }
        } else {
            const int addr = static_cast<int>(((lane * 5u + 3u) & 63u) * 4u);
#pragma unroll
            for (int i = 0; i < R; ++i) v[i] = static_cast<unsigned>(__builtin_amdgcn_ds_bpermute(addr, static_cast<int>(v[i])));
        }
    }
    unsigned acc = 0;
#pragma unroll
    for (int i = 0; i < R; ++i) acc ^= v[i] * (i + 1u);
    out[blockIdx.x * 64 + lane] = acc;
}

template <int MODE> void run(const char* name, unsigned* out, int blocks, int rounds)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(xchg<MODE>, dim3(blocks), dim3(64), 0, 0, out, rounds);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(xchg<MODE>, dim3(blocks), dim3(64), 0, 0, out, rounds);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    // each SIMD ran blocks / 1024 waves in turn-taking; cycles a SIMD spends per wave and exchange step:
    const double cyc = ms * 1e-3 * 2.4e9 / (double(blocks) / 1024.0 * rounds);
    printf("%-7s %8.3f ms  %7.1f SIMD cycles per step and wave\n", name, ms, cyc);
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
}

int main(int argc, char** argv)
{
    const int blocks = 1024 * 16, rounds = argc > 1 ? atoi(argv[1]) : 2000;
    unsigned* out; CK(hipMalloc(&out, size_t(blocks) * 64 * 4));
    run<0>("lds", out, blocks, rounds);
    run<1>("swap32", out, blocks, rounds);
    run<2>("swap16", out, blocks, rounds);
    run<3>("dpp8", out, blocks, rounds);
    run<4>("dpp1", out, blocks, rounds);
    run<5>("bperm", out, blocks, rounds);
    CK(hipFree(out));
    return 0;
}
