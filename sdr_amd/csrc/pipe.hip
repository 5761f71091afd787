// pipe.hip -- streaming ingest: a ring of slots that overlaps upload, the fused receive kernel
// and download on three streams (include/sdr_amd.h, "Streaming ingest").  The reference's
// analogue is the host ring utils::sliding_buffer (lib/inc/sliding_buffer.hpp:14-104) fed one
// OFDM symbol at a time by the demo (app/QFDemoWindow.cpp:213-229).
#include "common.hpp"
#include "kernels.hpp"

#include <mutex>
#include <vector>

using namespace sdr;

namespace {

struct Slot {
    void* h_rx = nullptr;
    void* d_rx = nullptr;
    float* d_llr = nullptr;
    uint8_t* d_hard = nullptr;
    float* h_llr = nullptr;
    uint8_t* h_hard = nullptr;
    hipEvent_t uploaded = nullptr, computed = nullptr, done = nullptr;
    uint64_t n_sym = 0;
};

} // namespace

struct sdr_pipe {
    int device = 0;
    sdr_pipe_config cfg = {};
    sdr_ctx* ctx = nullptr;                    // enqueues on `compute`
    hipStream_t up = nullptr, compute = nullptr, down = nullptr;
    std::vector<Slot> slots;
    std::mutex m;                              // guards the four counters below
    uint64_t submitted = 0, collected = 0, released = 0;
    bool acquired = false;                     // slot (submitted % depth) handed to the producer
};

namespace {

int use_device(sdr_pipe* p)
{
    if (!p) return fail(SDR_ERR_INVALID, "null pipe");
    SDR_HIP(hipSetDevice(p->device));
    return SDR_OK;
}

size_t rx_bytes(const sdr_pipe_config& c, uint64_t n_sym) { return size_t(n_sym) * (size_t(c.n_fft) + c.cp) * 8; }
size_t llr_bytes(const sdr_pipe_config& c, uint64_t n_sym) { return size_t(n_sym) * c.n_fft * c.mod * 4; }
size_t hard_bytes(const sdr_pipe_config& c, uint64_t n_sym) { return size_t(n_sym) * c.n_fft * c.mod / 8; }

} // namespace

extern "C" {

void sdr_pipe_destroy(sdr_pipe* p)
{
    if (!p) return;
    (void)hipSetDevice(p->device);
    for (hipStream_t s : {p->up, p->compute, p->down}) if (s) (void)hipStreamSynchronize(s);
    for (Slot& s : p->slots) {
        if (s.h_rx) (void)hipHostFree(s.h_rx);
        if (s.h_llr) (void)hipHostFree(s.h_llr);
        if (s.h_hard) (void)hipHostFree(s.h_hard);
        if (s.d_rx) (void)hipFree(s.d_rx);
        if (s.d_llr) (void)hipFree(s.d_llr);
        if (s.d_hard) (void)hipFree(s.d_hard);
        for (hipEvent_t e : {s.uploaded, s.computed, s.done}) if (e) (void)hipEventDestroy(e);
    }
    if (p->ctx) sdr_destroy(p->ctx);
    for (hipStream_t s : {p->up, p->compute, p->down}) if (s) (void)hipStreamDestroy(s);
    delete p;
}

int sdr_pipe_create(int device, const sdr_pipe_config* config, sdr_pipe** out)
{
    if (!config || !out) return fail(SDR_ERR_INVALID, "null argument");
    *out = nullptr;
    if (config->struct_size != sizeof(sdr_pipe_config)) return fail(SDR_ERR_INVALID, "sdr_pipe_config.struct_size %u != %zu", config->struct_size, sizeof(sdr_pipe_config));
    sdr_pipe_config c = *config;
    if (c.flags & SDR_PIPE_LLR_TO_HOST) c.flags |= SDR_PIPE_WANT_LLR;
    if (c.flags & SDR_PIPE_HARD_TO_HOST) c.flags |= SDR_PIPE_WANT_HARD;
    if (!(c.flags & (SDR_PIPE_WANT_LLR | SDR_PIPE_WANT_HARD))) return fail(SDR_ERR_INVALID, "pipe produces neither LLRs nor hard bits");
    if (c.depth < 2 || c.depth > 16) return fail(SDR_ERR_INVALID, "depth %u outside 2..16", c.depth);
    if (c.chunk_symbols == 0) return fail(SDR_ERR_INVALID, "chunk_symbols is 0");
    if (c.n_fft == 0 || (c.n_fft & (c.n_fft - 1)) != 0) return fail(SDR_ERR_NOT_POW2, "%s", kPow2Message);
    switch (c.mod) { case 1: case 2: case 4: case 6: case 8: case 10: break; default: return fail(SDR_ERR_INVALID, "unknown modulation %u", c.mod); }
    if ((size_t(c.chunk_symbols) * c.n_fft * c.mod) % 8 != 0) return fail(SDR_ERR_INVALID, "chunk_symbols*n_fft*mod must be a whole number of bytes");

    int n_dev = 0;
    SDR_HIP(hipGetDeviceCount(&n_dev));
    if (device < 0 || device >= n_dev) return fail(SDR_ERR_INVALID, "device %d of %d", device, n_dev);
    SDR_HIP(hipSetDevice(device));
    sdr_pipe* p = new sdr_pipe;
    p->device = device;
    p->cfg = c;
    auto bail = [&](int rc) { sdr_pipe_destroy(p); return rc; };
#define PIPE_HIP(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return bail(fail_hip(e_, #expr)); } while (0)
    PIPE_HIP(hipStreamCreateWithFlags(&p->up, hipStreamNonBlocking));
    PIPE_HIP(hipStreamCreateWithFlags(&p->compute, hipStreamNonBlocking));
    PIPE_HIP(hipStreamCreateWithFlags(&p->down, hipStreamNonBlocking));
    if (int rc = sdr_create_on_stream(device, p->compute, &p->ctx); rc != SDR_OK) return bail(rc);
    p->slots.resize(c.depth);
    for (Slot& s : p->slots) {
        PIPE_HIP(hipHostMalloc(&s.h_rx, rx_bytes(c, c.chunk_symbols), hipHostMallocDefault));
        PIPE_HIP(hipMalloc(&s.d_rx, rx_bytes(c, c.chunk_symbols)));
        if (c.flags & SDR_PIPE_WANT_LLR) PIPE_HIP(hipMalloc(reinterpret_cast<void**>(&s.d_llr), llr_bytes(c, c.chunk_symbols)));
        if (c.flags & SDR_PIPE_WANT_HARD) PIPE_HIP(hipMalloc(reinterpret_cast<void**>(&s.d_hard), hard_bytes(c, c.chunk_symbols)));
        if (c.flags & SDR_PIPE_LLR_TO_HOST) PIPE_HIP(hipHostMalloc(reinterpret_cast<void**>(&s.h_llr), llr_bytes(c, c.chunk_symbols), hipHostMallocDefault));
        if (c.flags & SDR_PIPE_HARD_TO_HOST) PIPE_HIP(hipHostMalloc(reinterpret_cast<void**>(&s.h_hard), hard_bytes(c, c.chunk_symbols), hipHostMallocDefault));
        PIPE_HIP(hipEventCreateWithFlags(&s.uploaded, hipEventDisableTiming));
        PIPE_HIP(hipEventCreateWithFlags(&s.computed, hipEventDisableTiming));
        PIPE_HIP(hipEventCreateWithFlags(&s.done, hipEventDisableTiming));
    }
#undef PIPE_HIP
    *out = p;
    return SDR_OK;
}

int sdr_pipe_acquire(sdr_pipe* p, void** h_rx)
{
    SDR_TRY(use_device(p));
    if (!h_rx) return fail(SDR_ERR_INVALID, "null h_rx");
    std::lock_guard<std::mutex> lock(p->m);
    if (p->submitted - p->released >= p->cfg.depth) return fail(SDR_ERR_INVALID, "all %u slots are in flight: collect and release first", p->cfg.depth);
    p->acquired = true;
    *h_rx = p->slots[p->submitted % p->cfg.depth].h_rx;
    return SDR_OK;
}

int sdr_pipe_submit(sdr_pipe* p, uint64_t n_sym)
{
    SDR_TRY(use_device(p));
    const sdr_pipe_config& c = p->cfg;
    Slot* s = nullptr;
    {
        std::lock_guard<std::mutex> lock(p->m);
        if (!p->acquired) return fail(SDR_ERR_INVALID, "submit without acquire");
        s = &p->slots[p->submitted % c.depth];
    }
    if (n_sym == 0 || n_sym > c.chunk_symbols) return fail(SDR_ERR_INVALID, "n_sym %llu outside 1..%llu", (unsigned long long)n_sym, (unsigned long long)c.chunk_symbols);
    if ((size_t(n_sym) * c.n_fft * c.mod) % 8 != 0) return fail(SDR_ERR_INVALID, "n_sym*n_fft*mod must be a whole number of bytes");
    s->n_sym = n_sym;
    SDR_HIP(hipMemcpyAsync(s->d_rx, s->h_rx, rx_bytes(c, n_sym), hipMemcpyHostToDevice, p->up));
    SDR_HIP(hipEventRecord(s->uploaded, p->up));
    SDR_HIP(hipStreamWaitEvent(p->compute, s->uploaded, 0));
    sdr_demod_args a = {};
    a.struct_size = sizeof(a);
    a.n_fft = c.n_fft; a.cp = c.cp; a.mod = c.mod; a.n_sym = n_sym;
    a.sym_per_set = c.sym_per_set; a.llr_weight0 = c.llr_weight0;
    a.d_rx = s->d_rx; a.d_eq = c.d_eq; a.d_llr_weight = c.d_llr_weight;
    a.d_llr = s->d_llr; a.d_hard = s->d_hard;
    SDR_TRY(sdr_ofdm_demod_c32(p->ctx, &a));
    SDR_HIP(hipEventRecord(s->computed, p->compute));
    SDR_HIP(hipStreamWaitEvent(p->down, s->computed, 0));
    if (s->h_llr) SDR_HIP(hipMemcpyAsync(s->h_llr, s->d_llr, llr_bytes(c, n_sym), hipMemcpyDeviceToHost, p->down));
    if (s->h_hard) SDR_HIP(hipMemcpyAsync(s->h_hard, s->d_hard, hard_bytes(c, n_sym), hipMemcpyDeviceToHost, p->down));
    SDR_HIP(hipEventRecord(s->done, p->down));
    std::lock_guard<std::mutex> lock(p->m);
    p->acquired = false;
    p->submitted += 1;
    return SDR_OK;
}

int sdr_pipe_collect(sdr_pipe* p, sdr_pipe_result* r)
{
    SDR_TRY(use_device(p));
    if (!r) return fail(SDR_ERR_INVALID, "null result");
    Slot* s = nullptr;
    uint64_t seq = 0;
    {
        std::lock_guard<std::mutex> lock(p->m);
        if (p->collected >= p->submitted) return fail(SDR_ERR_INVALID, "nothing submitted to collect");
        seq = p->collected;
        s = &p->slots[seq % p->cfg.depth];
    }
    SDR_HIP(hipEventSynchronize(s->done));
    r->sequence = seq;
    r->n_sym = s->n_sym;
    r->h_llr = s->h_llr;
    r->h_hard = s->h_hard;
    r->d_llr = s->d_llr;
    r->d_hard = s->d_hard;
    std::lock_guard<std::mutex> lock(p->m);
    p->collected += 1;
    return SDR_OK;
}

int sdr_pipe_release(sdr_pipe* p)
{
    SDR_TRY(use_device(p));
    std::lock_guard<std::mutex> lock(p->m);
    if (p->released >= p->collected) return fail(SDR_ERR_INVALID, "release without a collected chunk");
    p->released += 1;
    return SDR_OK;
}

int sdr_pipe_in_flight(sdr_pipe* p, uint32_t* count)
{
    if (!p || !count) return fail(SDR_ERR_INVALID, "null argument");
    std::lock_guard<std::mutex> lock(p->m);
    *count = static_cast<uint32_t>(p->submitted - p->released);
    return SDR_OK;
}

} // extern "C"
