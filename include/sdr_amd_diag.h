/* sdr_amd_diag.h -- measurement aids of libsdr_amd.so.  NOT part of the product ABI in sdr_amd.h:
 * nothing here has a reference counterpart or computes anything; benchmarks use it to quote the
 * receive kernel against what the memory system gives a copy of the same shape. */
#ifndef SDR_AMD_DIAG_H
#define SDR_AMD_DIAG_H

#include "sdr_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Moves exactly the bytes of sdr_ofdm_demod_c32 with LLR output only -- reads the n_sym*n_fft
 * payload samples (cyclic prefix skipped), writes n_sym*n_fft*mod floats (values unspecified) --
 * with the same access pattern and workgroup order, and no arithmetic. */
int sdr_diag_copy_shape_c32(sdr_ctx* ctx, const void* d_rx, uint32_t n_fft, uint32_t cp, uint32_t mod, uint64_t n_sym, float* d_llr);

/* Sets a run-time tuning knob of the receive kernels for every later launch of the process (the
 * environment variables SDR_LDS_PAD / SDR_XCD_WINDOW / SDR_DEBUG give the initial values):
 * "lds_pad" = unused dynamic LDS per workgroup in bytes (throttles resident workgroups per CU),
 * "xcd_window" = window of the XCD-aware workgroup order (0 = natural order), "debug" = 0/1. */
int sdr_diag_set_tuning(const char* name, long value);

#ifdef __cplusplus
}
#endif
#endif /* SDR_AMD_DIAG_H */
