// pipe.hpp -- C++ face of the streaming ingest ring (sdr_pipe_*, include/sdr_amd.h).
// Additive: the reference has no device and therefore no such type; what it has in this place
// is the host ring utils::sliding_buffer (lib/inc/sliding_buffer.hpp) fed one symbol per tick
// (app/QFDemoWindow.cpp:213-229).  Errors are values, like everywhere in the reference's API.
#pragma once

#include <complex>
#include <cstdint>
#include <span>
#include <string>
#include <utility>

#include "../sdr_amd.h"
#include "expected.hpp"

namespace sdrlib {

struct pipe_options {
    std::uint32_t n_fft = 1024;
    std::uint32_t cp = 256;
    std::uint32_t bits_per_symbol = 4;      // sdr_modulation
    std::uint64_t chunk_symbols = 4096;
    std::uint32_t depth = 3;
    bool llr_to_host = false;
    bool hard_to_host = true;
    bool keep_llr_on_device = false;
    float llr_weight = 1.0f;
    int device = 0;
};

struct pipe_chunk {
    std::uint64_t sequence = 0;
    std::uint64_t n_sym = 0;
    std::span<const float> llr;             // empty unless llr_to_host
    std::span<const std::uint8_t> hard;     // empty unless hard_to_host
    const float* d_llr = nullptr;
    const std::uint8_t* d_hard = nullptr;
};

class pipe {
public:
    static expected<pipe, std::string> create(const pipe_options& o)
    {
        sdr_pipe_config c = {};
        c.struct_size = sizeof(c);
        c.n_fft = o.n_fft; c.cp = o.cp; c.mod = o.bits_per_symbol;
        c.chunk_symbols = o.chunk_symbols; c.depth = o.depth;
        c.flags = (o.llr_to_host ? SDR_PIPE_LLR_TO_HOST : 0) | (o.hard_to_host ? SDR_PIPE_HARD_TO_HOST : 0) | (o.keep_llr_on_device ? SDR_PIPE_WANT_LLR : 0);
        c.llr_weight0 = o.llr_weight;
        sdr_pipe* h = nullptr;
        if (sdr_pipe_create(o.device, &c, &h) != SDR_OK) return unexpected<std::string>(sdr_last_error());
        return pipe(h, o);
    }

    pipe(pipe&& other) noexcept : handle_(std::exchange(other.handle_, nullptr)), opt_(other.opt_) {}
    pipe& operator=(pipe&& other) noexcept
    {
        if (this != &other) { reset(); handle_ = std::exchange(other.handle_, nullptr); opt_ = other.opt_; }
        return *this;
    }
    pipe(const pipe&) = delete;
    pipe& operator=(const pipe&) = delete;
    ~pipe() { reset(); }

    // Pinned input of the next free slot: chunk_symbols * (n_fft + cp) samples.
    expected<std::span<std::complex<float>>, std::string> acquire()
    {
        void* p = nullptr;
        if (sdr_pipe_acquire(handle_, &p) != SDR_OK) return unexpected<std::string>(sdr_last_error());
        return std::span<std::complex<float>>(static_cast<std::complex<float>*>(p), opt_.chunk_symbols * (std::size_t(opt_.n_fft) + opt_.cp));
    }
    expected<void, std::string> submit(std::uint64_t n_sym) { return status(sdr_pipe_submit(handle_, n_sym)); }
    expected<pipe_chunk, std::string> collect()
    {
        sdr_pipe_result r = {};
        if (sdr_pipe_collect(handle_, &r) != SDR_OK) return unexpected<std::string>(sdr_last_error());
        pipe_chunk c;
        c.sequence = r.sequence; c.n_sym = r.n_sym; c.d_llr = r.d_llr; c.d_hard = r.d_hard;
        const std::size_t bits = std::size_t(r.n_sym) * opt_.n_fft * opt_.bits_per_symbol;
        if (r.h_llr) c.llr = std::span<const float>(r.h_llr, bits);
        if (r.h_hard) c.hard = std::span<const std::uint8_t>(r.h_hard, bits / 8);
        return c;
    }
    expected<void, std::string> release() { return status(sdr_pipe_release(handle_)); }
    std::uint32_t in_flight() const { std::uint32_t n = 0; (void)sdr_pipe_in_flight(handle_, &n); return n; }
    const pipe_options& options() const { return opt_; }

private:
    pipe(sdr_pipe* h, const pipe_options& o) : handle_(h), opt_(o) {}
    void reset() { if (handle_) { sdr_pipe_destroy(handle_); handle_ = nullptr; } }
    static expected<void, std::string> status(int rc)
    {
        if (rc != SDR_OK) return unexpected<std::string>(sdr_last_error());
        return {};
    }
    sdr_pipe* handle_ = nullptr;
    pipe_options opt_;
};

} // namespace sdrlib
