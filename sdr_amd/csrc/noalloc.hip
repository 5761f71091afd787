// noalloc.hip -- sdr_ofdm_demod_noalloc_c32: the fused receive path for callers that cannot
// allocate (HIP graph capture, tight serving loops, harnesses that forbid hipMalloc).
//
// The existing entry point sdr_ofdm_demod_c32 uses two pieces of device memory beyond its inputs
// and outputs, both allocated lazily by the context: the forward twiddle table of the FFT size
// (n_fft - 1 complex values) and the 1024 x 8 counter slots that the counting kernels add into
// before one fold kernel sums them into the caller's eight.  Here both live in ONE file-scope
// device array, `scratch`, 192 KiB per GPU, held until the process ends:
//   [0, 64 KiB)        kCounterSlots x 8 uint64 counter slots
//   [64 KiB, 192 KiB)  the forward twiddle tables of every tuned size, n_fft = 64 ... 8192, table of
//                      2^L at byte offset 64 KiB + 8 * (2^L - 64)   (2^L - 1 entries each)
// The array's name is only a host-side stand-in: its device address is resolved once per device
// with hipGetSymbolAddress (noalloc_setup_device, called from sdr_create with that device
// current) and kept in g_dev[device].ptr; every use below goes through that pointer.  The tables
// are uploaded there once, at the same set-up.  The entry point itself issues one hipMemsetAsync
// (the slots must start at zero), the receive kernel and the fold kernel -- no hipMalloc, no
// hipFree, no copy.  All calls on one GPU share that GPU's array: calls that overlap in time on
// one GPU (two contexts, two streams) race on the slots.
#include "common.hpp"
#include "kernels.hpp"
#include "qam.hpp"

#include <mutex>
#include <vector>

namespace sdr {

namespace {

constexpr size_t kSlotBytes = static_cast<size_t>(kCounterSlots) * 8 * sizeof(unsigned long long);   // 65536
constexpr unsigned kMinLog2 = 6, kMaxLog2 = 13;                                                     // tuned sizes
constexpr size_t kTableBytes = ((size_t(1) << (kMaxLog2 + 1)) - (size_t(1) << kMinLog2)) * 8;        // 130560
constexpr size_t kScratchBytes = kSlotBytes + ((kTableBytes + 255) & ~size_t(255));                 // 196096

static __device__ __align__(256) __attribute__((used)) unsigned char scratch[kScratchBytes];

constexpr size_t table_offset(unsigned log2n) { return kSlotBytes + ((size_t(1) << log2n) - (size_t(1) << kMinLog2)) * 8; }

constexpr int kMaxDevices = 64;
struct DeviceScratch {
    unsigned char* ptr = nullptr;      // device address of this GPU's copy of `scratch`
    bool ready = false;
};
DeviceScratch g_dev[kMaxDevices];
std::mutex g_mutex;

} // namespace

// Called with `device` current (sdr_create).  Once per device: resolve the array's address on that
// device, clear the slots, upload the eight twiddle tables.
int noalloc_setup_device(int device)
{
    if (device < 0 || device >= kMaxDevices) return fail(SDR_ERR_INVALID, "device %d beyond the %d the allocation-free path keeps state for", device, kMaxDevices);
    std::lock_guard<std::mutex> lock(g_mutex);
    DeviceScratch& d = g_dev[device];
    if (d.ready) return SDR_OK;
    unsigned char* ptr = nullptr;
    hipError_t e = hipGetSymbolAddress(reinterpret_cast<void**>(&ptr), HIP_SYMBOL(scratch));
    if (e != hipSuccess) return fail_hip(e, "hipGetSymbolAddress(scratch)");
    if (!ptr) return fail(SDR_ERR_HIP, "hipGetSymbolAddress(scratch) returned a null address");
    std::vector<float> host(kTableBytes / sizeof(float), 0.0f);
    for (unsigned l = kMinLog2; l <= kMaxLog2; ++l)
        make_twiddles_f32(size_t(1) << l, false, host.data() + (table_offset(l) - kSlotBytes) / sizeof(float));
    SDR_HIP(hipMemset(ptr, 0, kSlotBytes));
    SDR_HIP(hipMemcpy(ptr + kSlotBytes, host.data(), kTableBytes, hipMemcpyHostToDevice));     // synchronous: `host` dies at return
    d.ptr = ptr;
    d.ready = true;
    return SDR_OK;
}

} // namespace sdr

using namespace sdr;

extern "C" int sdr_ofdm_demod_noalloc_c32(sdr_ctx* ctx, const sdr_demod_args* args)
{
    SDR_TRY(ensure_device(ctx));
    if (!args) return fail(SDR_ERR_INVALID, "null args");
    if (args->struct_size != sizeof(sdr_demod_args)) return fail(SDR_ERR_INVALID, "sdr_demod_args.struct_size %u != %zu", args->struct_size, sizeof(sdr_demod_args));
    const sdr_demod_args& a = *args;
    if (!valid_mod(static_cast<int>(a.mod))) return fail(SDR_ERR_INVALID, "unknown modulation %u", a.mod);
    if (a.n_sym == 0 || a.n_fft == 0) return SDR_OK;
    if (!is_pow2(a.n_fft)) return fail(SDR_ERR_NOT_POW2, "%s", kPow2Message);
    if (!a.d_rx) return fail(SDR_ERR_INVALID, "null d_rx");
    if ((a.d_ref_hard != nullptr) != (a.d_counters != nullptr)) return fail(SDR_ERR_INVALID, "d_ref_hard and d_counters must be given together");
    // the static array holds tables for n_fft = 64 ... 8192 only, and the generic engines behind the
    // other sizes and layouts need staging memory this entry point must not allocate
    if (a.n_fft < (1u << kMinLog2) || a.n_fft > (1u << kMaxLog2))
        return fail(SDR_ERR_UNSUPPORTED, "sdr_ofdm_demod_noalloc_c32 covers n_fft = %u ... %u (its static tables), got %u; use sdr_ofdm_demod_c32", 1u << kMinLog2, 1u << kMaxLog2, a.n_fft);
    if (!rx_fused_supported(a.n_fft, a.mod) || !rx_fused_layout_ok(a))
        return fail(SDR_ERR_UNSUPPORTED, "sdr_ofdm_demod_noalloc_c32 needs an even cp and 16-byte aligned buffers (the other layouts stage through allocated memory); use sdr_ofdm_demod_c32");
    if (ctx->device < 0 || ctx->device >= kMaxDevices || !g_dev[ctx->device].ready)
        return fail(SDR_ERR_INVALID, "the static scratch array of device %d was never set up (context not made by sdr_create?)", ctx->device);
    unsigned char* base = g_dev[ctx->device].ptr;          // never the array's name: that is not a device address
    RxWorkspace ws;
    ws.twiddles = base + table_offset(ilog2(a.n_fft));
    ws.counter_slots = reinterpret_cast<unsigned long long*>(base);
    // the slots keep whatever the last call on this GPU left (zero after a completed fold, anything
    // after a failed launch): the counting kernels need them at zero
    if (a.d_counters) SDR_HIP(hipMemsetAsync(base, 0, kSlotBytes, ctx->stream));
    return rx_fused_launch(ctx, a, &ws);
}
