// noalloc.hip -- sdr_ofdm_demod_noalloc_c32: the fused receive path for callers that cannot
// allocate (HIP graph capture, tight serving loops, harnesses that forbid hipMalloc).
//
// The existing entry point sdr_ofdm_demod_c32 uses two pieces of device memory beyond its inputs
// and outputs, both allocated lazily by the context: the forward twiddle table of the FFT size
// (n_fft - 1 complex values) and the 1024 x 8 counter slots that the counting kernels add into
// before one fold kernel sums them into the caller's eight.  Here both live in ONE file-scope
// device array, `scratch`, 384 KiB per GPU, held until the process ends:
//   [0, 256 KiB)        kSlotSets = 4 sets of kCounterSlots x 8 uint64 counter slots; a context claims one
//                       set at sdr_create and gives it back at sdr_destroy, so that contexts (each with
//                       its own stream) never share slots
//   [256 KiB, 384 KiB)  the forward twiddle tables of every tuned size, n_fft = 64 ... 8192, table of
//                       2^L at byte offset 256 KiB + 8 * (2^L - 64)   (2^L - 1 entries each)
// The array's name is only a host-side stand-in: its device address is resolved once per device
// with hipGetSymbolAddress (noalloc_attach, called from sdr_create with that device current) and
// kept per device and per context; every use below goes through that pointer.  The tables are
// uploaded there once, at the same set-up, on a short-lived stream of their own (never the legacy
// stream).  A failed set-up does not fail sdr_create: it is recorded per device and reported by
// this entry point only.  The entry point itself issues one hipMemsetAsync (the slots must start
// at zero), the receive kernel and the fold kernel -- no hipMalloc, no hipFree, no copy.  A fifth
// live context on one GPU gets no slot set: its LLR-only / hard-only calls work, its counting
// calls are refused (SDR_ERR_UNSUPPORTED) instead of racing with another context's.
#include "common.hpp"
#include "kernels.hpp"
#include "qam.hpp"

#include <atomic>
#include <mutex>
#include <string>
#include <vector>

namespace sdr {

namespace {

constexpr size_t kSlotBytes = static_cast<size_t>(kCounterSlots) * 8 * sizeof(unsigned long long);   // 65536
constexpr int kSlotSets = 4;
constexpr size_t kAllSlotBytes = kSlotBytes * kSlotSets;                                            // 262144
constexpr unsigned kMinLog2 = 6, kMaxLog2 = 13;                                                     // tuned sizes
constexpr size_t kTableBytes = ((size_t(1) << (kMaxLog2 + 1)) - (size_t(1) << kMinLog2)) * 8;        // 130560
constexpr size_t kScratchBytes = kAllSlotBytes + ((kTableBytes + 255) & ~size_t(255));              // 392704

static __device__ __align__(256) __attribute__((used)) unsigned char scratch[kScratchBytes];

constexpr size_t table_offset(unsigned log2n) { return kAllSlotBytes + ((size_t(1) << log2n) - (size_t(1) << kMinLog2)) * 8; }

constexpr int kMaxDevices = 64;
struct DeviceScratch {
    unsigned char* ptr = nullptr;      // device address of this GPU's copy of `scratch`; written before `ready` is released
    std::atomic<bool> ready{false};
    bool tried = false;                // under g_mutex
    int error_code = SDR_OK;           // under g_mutex: why the set-up failed
    std::string error_text;
    unsigned sets_in_use = 0;          // under g_mutex: bit i = slot set i belongs to a live context
};
DeviceScratch g_dev[kMaxDevices];
std::mutex g_mutex;

// once per device, with that device current and g_mutex held
int setup_device_locked(DeviceScratch& d)
{
    unsigned char* ptr = nullptr;
    hipError_t e = hipGetSymbolAddress(reinterpret_cast<void**>(&ptr), HIP_SYMBOL(scratch));
    if (e != hipSuccess) return fail_hip(e, "hipGetSymbolAddress(scratch)");
    if (!ptr) return fail(SDR_ERR_HIP, "hipGetSymbolAddress(scratch) returned a null address");
    std::vector<float> host(kTableBytes / sizeof(float), 0.0f);
    for (unsigned l = kMinLog2; l <= kMaxLog2; ++l)
        make_twiddles_f32(size_t(1) << l, false, host.data() + (table_offset(l) - kAllSlotBytes) / sizeof(float));
    // a stream of its own: the legacy stream would synchronise with (and, under another thread's
    // global-mode capture, invalidate) work this set-up has nothing to do with
    hipStream_t s = nullptr;
    SDR_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    e = hipMemsetAsync(ptr, 0, kAllSlotBytes, s);
    if (e == hipSuccess) e = hipMemcpyAsync(ptr + kAllSlotBytes, host.data(), kTableBytes, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);          // `host` dies at return
    (void)hipStreamDestroy(s);
    if (e != hipSuccess) return fail_hip(e, "upload of the static twiddle tables");
    d.ptr = ptr;
    d.ready.store(true, std::memory_order_release);
    return SDR_OK;
}

} // namespace

// Called by sdr_create with ctx->device current.  Never fails the creation: a set-up error is kept
// per device and comes back from sdr_ofdm_demod_noalloc_c32.
void noalloc_attach(sdr_ctx* ctx)
{
    ctx->noalloc_base = nullptr;
    ctx->noalloc_slot_set = -1;
    if (ctx->device < 0 || ctx->device >= kMaxDevices) return;
    std::lock_guard<std::mutex> lock(g_mutex);
    DeviceScratch& d = g_dev[ctx->device];
    if (!d.tried) {
        d.tried = true;
        d.error_code = setup_device_locked(d);
        if (d.error_code != SDR_OK) d.error_text = sdr_last_error();
    }
    if (!d.ready.load(std::memory_order_acquire)) return;
    ctx->noalloc_base = d.ptr;
    for (int i = 0; i < kSlotSets; ++i) {
        if (!(d.sets_in_use & (1u << i))) {
            d.sets_in_use |= 1u << i;
            ctx->noalloc_slot_set = i;
            break;
        }
    }
}

void noalloc_detach(sdr_ctx* ctx)
{
    if (ctx->noalloc_slot_set < 0 || ctx->device < 0 || ctx->device >= kMaxDevices) return;
    std::lock_guard<std::mutex> lock(g_mutex);
    g_dev[ctx->device].sets_in_use &= ~(1u << ctx->noalloc_slot_set);
    ctx->noalloc_slot_set = -1;
}

} // namespace sdr

using namespace sdr;

extern "C" int sdr_ofdm_demod_noalloc_c32(sdr_ctx* ctx, const sdr_demod_args* args)
{
    SDR_TRY(ensure_device(ctx));
    sdr_demod_args a;
    SDR_TRY(read_demod_args(args, a));
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
    unsigned char* base = ctx->noalloc_base;               // never the array's name: that is not a device address
    if (!base) {
        if (ctx->device < 0 || ctx->device >= kMaxDevices) return fail(SDR_ERR_UNSUPPORTED, "device %d beyond the %d the allocation-free path keeps state for", ctx->device, kMaxDevices);
        std::lock_guard<std::mutex> lock(g_mutex);
        const DeviceScratch& d = g_dev[ctx->device];
        if (d.error_code != SDR_OK) return fail(static_cast<sdr_status>(d.error_code), "static scratch of device %d was not set up: %s", ctx->device, d.error_text.c_str());
        return fail(SDR_ERR_INVALID, "the static scratch array of device %d was never set up (context not made by sdr_create?)", ctx->device);
    }
    RxWorkspace ws;
    ws.twiddles = base + table_offset(ilog2(a.n_fft));
    if (a.d_counters) {
        if (ctx->noalloc_slot_set < 0)
            return fail(SDR_ERR_UNSUPPORTED, "counting without allocation: all %d static counter-slot sets of device %d belong to other live contexts; destroy one or use sdr_ofdm_demod_c32", kSlotSets, ctx->device);
        unsigned char* slots = base + static_cast<size_t>(ctx->noalloc_slot_set) * kSlotBytes;
        ws.counter_slots = reinterpret_cast<unsigned long long*>(slots);
        // the slots keep whatever the context's last call left (zero after a completed fold, anything
        // after a failed launch): the counting kernels need them at zero
        SDR_HIP(hipMemsetAsync(slots, 0, kSlotBytes, ctx->stream));
    }
    return rx_fused_launch(ctx, a, &ws);
}
