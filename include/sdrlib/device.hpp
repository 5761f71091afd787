// device.hpp -- the device context the drop-in headers run on.
//
// The reference's functions are stateless free templates (lib/inc/*.hpp); here each calling
// thread lazily owns one sdr_ctx (one HIP stream on device SDRLIB_DEVICE, default 0), so calls
// from different threads stay independent, as in the reference.  There is no CPU fallback: if
// no context can be created the error text of libsdr_amd is returned through std::expected.
#pragma once

#include <cstdlib>
#include <string>

#include "../sdr_amd.h"

namespace sdrlib {

class thread_context {
public:
    // Returns the calling thread's context, or nullptr with `why` set.
    static sdr_ctx* get(std::string& why)
    {
        thread_local thread_context self;
        if (!self.ctx_) {
            int dev = 0;
            if (const char* e = std::getenv("SDRLIB_DEVICE")) dev = std::atoi(e);
            if (sdr_create(dev, &self.ctx_) != SDR_OK) {
                why = sdr_last_error();
                self.ctx_ = nullptr;
            }
        }
        return self.ctx_;
    }
    thread_context(const thread_context&) = delete;
    thread_context& operator=(const thread_context&) = delete;

private:
    thread_context() = default;
    ~thread_context() { if (ctx_) sdr_destroy(ctx_); }
    sdr_ctx* ctx_ = nullptr;
};

} // namespace sdrlib
