// common.hpp -- context, error reporting and launch helpers shared by the HIP sources.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <map>
#include <tuple>
#include <vector>

#include "sdr_amd.h"

namespace sdr {

constexpr const char* kPow2Message = "The sequence size must be of powers of 2";   // reference lib/inc/fft.hpp:92

void set_error(const char* fmt, ...) __attribute__((format(printf, 1, 2)));
int fail(sdr_status code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
int fail_hip(hipError_t err, const char* what);

#define SDR_HIP(expr)                                            \
    do {                                                         \
        hipError_t sdr_hip_err_ = (expr);                        \
        if (sdr_hip_err_ != hipSuccess) return ::sdr::fail_hip(sdr_hip_err_, #expr); \
    } while (0)

#define SDR_TRY(expr)                 \
    do {                              \
        int sdr_rc_ = (expr);         \
        if (sdr_rc_ != SDR_OK) return sdr_rc_; \
    } while (0)

inline bool is_pow2(size_t n) { return n != 0 && (n & (n - 1)) == 0; }
inline unsigned ilog2(size_t n) { unsigned l = 0; while ((size_t(1) << l) < n) ++l; return l; }

struct Staging {
    void* ptr = nullptr;
    size_t bytes = 0;
};

} // namespace sdr

// The opaque context of the C ABI: one device, one stream, cached twiddle tables, staging.
struct sdr_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int num_cu = 0;
    size_t lds_max = 0;   // bytes of LDS one workgroup may use
    // key: (n, inverse, is_double) -> device table of n-1 complex twiddles
    std::map<std::tuple<size_t, int, int>, void*> twiddles;
    sdr::Staging stage[10];
    // pinned host buffer for small tables that kernels read (sweep point table): filled on the host,
    // copied asynchronously; pinned_done = the last copy out of it (waited for before it is rewritten)
    void* pinned = nullptr;
    size_t pinned_bytes = 0;
    hipEvent_t pinned_done = nullptr;
    // second stream and hand-off events of the chunked link (link_chain): the generator of chunk i+1
    // runs beside the demodulation of chunk i
    hipStream_t aux_stream = nullptr;
    hipEvent_t chain_generated[2] = {nullptr, nullptr};
    hipEvent_t chain_consumed[2] = {nullptr, nullptr};
    // kCounterSlots x 8 partial counters; kernels add into slot (workgroup id mod kCounterSlots),
    // fold_counters() sums them into the caller's eight and leaves them zero again.
    unsigned long long* counter_slots = nullptr;
    // host copy of the point table of the last sweep and where its device copy lies (link_sweep reuses it)
    std::vector<unsigned char> sweep_table;
    void* sweep_table_dev = nullptr;
    // tables of the Gaussian generator (rng.hpp), built on first use
    void* rng_tables = nullptr;
    // sdr_ofdm_demod_noalloc_c32 (noalloc.hip): this device's static scratch array and the counter-slot
    // set inside it that this context owns (-1: none free when the context was made)
    unsigned char* noalloc_base = nullptr;
    int noalloc_slot_set = -1;
};

namespace sdr {

int ensure_device(sdr_ctx* ctx);
// Device pointer to the reference-recurrence twiddle table for length n (n >= 2, power of two).
int get_twiddles(sdr_ctx* ctx, size_t n, bool inverse, bool is_double, const void** d_table);
int get_staging(sdr_ctx* ctx, int slot, size_t bytes, void** ptr);
// Copies `bytes` of host data into staging slot `slot` on the context's stream without blocking it
// (through the context's pinned buffer); *d_ptr = the device copy.
int upload_table(sdr_ctx* ctx, const void* src, size_t bytes, int slot, void** d_ptr);
constexpr int kCounterSlots = 1024;
int get_counter_slots(sdr_ctx* ctx, unsigned long long** slots);
// adds the slots and the counts that need no measuring (bits, symbols, OFDM symbols) into d_counters
int fold_counters(sdr_ctx* ctx, unsigned long long* d_counters, uint64_t n_points, uint32_t bps, uint64_t n_ofdm);
// the same on a caller-provided slot array (no allocation)
int fold_counter_slots(sdr_ctx* ctx, unsigned long long* slots, unsigned long long* d_counters, uint64_t n_points, uint32_t bps, uint64_t n_ofdm);

// Host-side generation of the reference's per-stage twiddle sequence (lib/inc/fft.hpp:113-150).
void make_twiddles_f32(size_t n, bool inverse, float* out);
void make_twiddles_f64(size_t n, bool inverse, double* out);
// Tables of the Gaussian generator (rng.hpp): 2 * (32 + 256) floats on the host / cached on the device.
void make_rng_tables_f32(float* out);
int get_rng_tables(sdr_ctx* ctx, const float2** d_tables);

} // namespace sdr
