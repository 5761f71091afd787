/* sdr_amd.h -- C ABI of libsdr_amd.so: the MI355X (gfx950) implementation of the erithion/SDR
 * receive-side hot path (CP removal -> batched FFT -> one-tap equalisation -> max-log QAM
 * soft demapping), plus the transmit chain / AWGN frame generator / BER counters around it.
 *
 * The reference (header-only C++, /root/reference/lib/inc) has no ABI of its own; each entry
 * point below names the reference interface it stands behind (file:line).  The drop-in C++
 * headers in include/sdrlib/ keep the reference's own names and call these.
 *
 * Conventions
 *  - Complex data is interleaved (re, im).  "c32" = two fp32 (std::complex<float>), "c64" = two
 *    fp64 (std::complex<double>).
 *  - Every function returns SDR_OK (0) or a negative sdr_status; sdr_last_error() returns the
 *    message of the calling thread's last failure.  A size that is not a power of two yields
 *    SDR_ERR_NOT_POW2 with exactly the reference's text
 *    "The sequence size must be of powers of 2" (lib/inc/fft.hpp:92).
 *  - A context owns one device and one HIP stream.  All work is enqueued on that stream and is
 *    asynchronous unless the name ends in _host (those copy in, run, copy out, and synchronise).
 *    One context must not be used from two threads at once; separate contexts are independent.
 *  - The library never frees or reallocates caller memory.  Device pointers must belong to the
 *    context's device.
 *  - There is NO CPU fallback: without a HIP device every compute entry point fails.
 */
#ifndef SDR_AMD_H
#define SDR_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct sdr_ctx sdr_ctx;

typedef enum sdr_status {
    SDR_OK = 0,
    SDR_ERR_NOT_POW2 = -1,   /* lib/inc/fft.hpp:91-92 */
    SDR_ERR_INVALID = -2,    /* bad argument (null pointer, cp > input, unknown modulation ...) */
    SDR_ERR_HIP = -3,        /* a HIP runtime call failed; text carries hipGetErrorString */
    SDR_ERR_UNSUPPORTED = -4 /* size/mode outside what the kernels cover */
} sdr_status;

/* Modulation = bits per subcarrier.  e16QAM is the reference's (lib/inc/modulation.hpp:15-68);
 * the others fill its empty tag structs (:13-14, :70-72) by the same rule: high half of the
 * index -> I, low half -> Q, per-axis Gray, level = 2*gray_decode(g)-(L-1), norm = 1/sqrt(Es). */
typedef enum sdr_modulation {
    SDR_MOD_BPSK = 1,
    SDR_MOD_QPSK = 2,
    SDR_MOD_16QAM = 4,
    SDR_MOD_64QAM = 6,
    SDR_MOD_256QAM = 8,
    SDR_MOD_1024QAM = 10
} sdr_modulation;

/* ---- library / context ------------------------------------------------------------------ */
const char* sdr_version(void);
const char* sdr_last_error(void);
int sdr_device_count(int* count);

/* Creates a context on `device` with a stream of its own. */
int sdr_create(int device, sdr_ctx** out);
/* Same, but enqueues on a caller-owned hipStream_t (e.g. torch's current stream). */
int sdr_create_on_stream(int device, void* hip_stream, sdr_ctx** out);
void sdr_destroy(sdr_ctx* ctx);
int sdr_synchronize(sdr_ctx* ctx);
/* hipStream_t the context enqueues on (for event timing by the caller). */
void* sdr_stream(sdr_ctx* ctx);

/* ---- R2/R4: fft::fft2 / fft::ifft2 (lib/inc/fft.hpp:176-180, :194-205) --------------------
 * In-place, batched: `batch` contiguous sequences of `n` complex values in device memory.
 * Same radix-2 DIT network and the same per-stage twiddle sequence as the reference
 * (:87-155); inverse also divides by n (:200-202).  n = 0 or 1 succeeds and does nothing (:91). */
int sdr_fft2_c32(sdr_ctx* ctx, void* d_data, size_t n, size_t batch, int inverse);
int sdr_fft2_c64(sdr_ctx* ctx, void* d_data, size_t n, size_t batch, int inverse);
/* Host-memory forms used by the drop-in fft.hpp. */
int sdr_fft2_host_c32(sdr_ctx* ctx, void* h_data, size_t n, size_t batch, int inverse);
int sdr_fft2_host_c64(sdr_ctx* ctx, void* h_data, size_t n, size_t batch, int inverse);
/* The per-stage twiddle table the kernels use: entry (span/2-1+j) = w of butterfly j of the
 * stage with that span (lib/inc/fft.hpp:113-150), n-1 complex entries, written to host memory. */
int sdr_twiddle_table_c32(size_t n, int inverse, float* h_out);
int sdr_twiddle_table_c64(size_t n, int inverse, double* h_out);

/* ---- R1/R5: ofdm::rx / ofdm::tx (lib/inc/ofdm.hpp:37-43, :13-24), batched ------------------
 * rx: in = n_sym x (n_fft+cp) samples, out = n_sym x n_fft bins (CP skipped on load).
 * tx: in = n_sym x n_fft bins, out = n_sym x (n_fft+cp) samples (CP = copy of the tail).
 * The batched C ABI requires cp <= n_fft for tx (SDR_ERR_INVALID otherwise); the reference's
 * sequence for cp > n (lib/inc/ofdm.hpp:16-21 copies past the payload) is kept only by the
 * one-symbol drop-in ofdm::tx in include/sdrlib/ofdm.hpp.  rx accepts any cp. */
int sdr_ofdm_rx_c32(sdr_ctx* ctx, const void* d_in, size_t n_fft, size_t cp, size_t n_sym, void* d_out);
int sdr_ofdm_rx_c64(sdr_ctx* ctx, const void* d_in, size_t n_fft, size_t cp, size_t n_sym, void* d_out);
int sdr_ofdm_tx_c32(sdr_ctx* ctx, const void* d_in, size_t n_fft, size_t cp, size_t n_sym, void* d_out);
int sdr_ofdm_tx_c64(sdr_ctx* ctx, const void* d_in, size_t n_fft, size_t cp, size_t n_sym, void* d_out);
int sdr_ofdm_rx_host_c32(sdr_ctx* ctx, const void* h_in, size_t n_fft, size_t cp, size_t n_sym, void* h_out);
int sdr_ofdm_rx_host_c64(sdr_ctx* ctx, const void* h_in, size_t n_fft, size_t cp, size_t n_sym, void* h_out);
int sdr_ofdm_tx_host_c32(sdr_ctx* ctx, const void* h_in, size_t n_fft, size_t cp, size_t n_sym, void* h_out);
int sdr_ofdm_tx_host_c64(sdr_ctx* ctx, const void* h_in, size_t n_fft, size_t cp, size_t n_sym, void* h_out);

/* ---- R6-R8: modulation::to_constl / from_constl / nearest (lib/inc/modulation.hpp:49-112) --
 * Bit streams are MSB first in groups of `mod` bits (for 16-QAM: high nibble first, :87-91).
 * to_constl: n_sym symbols from ceil(n_sym*mod/8) bytes.  from_constl: floor(n_sym*mod/8)
 * bytes of hard decisions (symbols that do not complete a byte are dropped, :103);
 * decisions follow nearest(): scale by inorm, ascending scan, strict '<' (:52-65). */
int sdr_to_constl_c32(sdr_ctx* ctx, int mod, const uint8_t* d_bytes, size_t n_sym, void* d_out);
int sdr_to_constl_c64(sdr_ctx* ctx, int mod, const uint8_t* d_bytes, size_t n_sym, void* d_out);
int sdr_from_constl_c32(sdr_ctx* ctx, int mod, const void* d_sym, size_t n_sym, uint8_t* d_bytes);
int sdr_from_constl_c64(sdr_ctx* ctx, int mod, const void* d_sym, size_t n_sym, uint8_t* d_bytes);
int sdr_to_constl_host_c32(sdr_ctx* ctx, int mod, const uint8_t* h_bytes, size_t n_sym, void* h_out);
int sdr_to_constl_host_c64(sdr_ctx* ctx, int mod, const uint8_t* h_bytes, size_t n_sym, void* h_out);
int sdr_from_constl_host_c32(sdr_ctx* ctx, int mod, const void* h_sym, size_t n_sym, uint8_t* h_bytes);
int sdr_from_constl_host_c64(sdr_ctx* ctx, int mod, const void* h_sym, size_t n_sym, uint8_t* h_bytes);
/* nearest() per symbol: one uint16 index each. */
int sdr_nearest_c32(sdr_ctx* ctx, int mod, const void* d_sym, size_t n_sym, uint16_t* d_index);
int sdr_nearest_c64(sdr_ctx* ctx, int mod, const void* d_sym, size_t n_sym, uint16_t* d_index);
int sdr_nearest_host_c32(sdr_ctx* ctx, int mod, const void* h_sym, size_t n_sym, uint16_t* h_index);
int sdr_nearest_host_c64(sdr_ctx* ctx, int mod, const void* h_sym, size_t n_sym, uint16_t* h_index);
/* Host-side description of a constellation: norm, inorm, 2^mod (I,Q) levels. */
int sdr_constellation_f32(int mod, float* norm, float* inorm, float* table);

/* ---- X1/X2: the fused receive path (no reference counterpart; sits between ofdm::rx,
 * lib/inc/ofdm.hpp:37-43, and modulation::from_constl, lib/inc/modulation.hpp:96-112) -------
 * Per OFDM symbol: skip cp samples, FFT (reference network), yhat = y*G_k, u = yhat*inorm,
 *   LLR_b = (min_{levels with bit 1}(v-l)^2 - min_{levels with bit 0}(v-l)^2) * w_k
 * (v = the I or Q coordinate of u that carries bit b; positive = bit 0), and the nearest()
 * hard decision.  Outputs that are NULL are skipped.
 *   d_llr   n_sym*n_fft*mod fp32, subcarrier-major, bit order = index MSB first (I bits, then Q).
 *   d_hard  n_sym*n_fft*mod/8 bytes, MSB-first stream over the whole batch.
 *   d_fft   n_sym*n_fft c32: the un-equalised FFT output (= ofdm::rx).
 * d_eq / d_llr_weight hold one set of n_fft taps / weights per `sym_per_set` consecutive
 * symbols (sym_per_set = 0: a single set); NULL d_eq = no equaliser, NULL d_llr_weight = the
 * scalar llr_weight0 -- or, with SDR_DEMOD_WEIGHT_FROM_TAPS in `flags` and taps given, the weight that belongs
 * to a zero-forcing tap, w_k = llr_weight0 / |G_k|^2 (= norm^2 |H_k|^2 / N0 when llr_weight0 = norm^2 / N0 and
 * G_k = conj(H_k) / |H_k|^2), formed in the kernel from the tap it already holds: a third fewer tap bytes and half
 * the tap loads of the two-array form.  If d_ref_hard (same layout as d_hard) and d_counters are given, the
 * kernel also accumulates into d_counters[0..7] (uint64):
 *   [0] bit errors  [1] bits  [2] subcarrier-symbol errors  [3] subcarrier symbols
 *   [4] OFDM symbols  [5] sum round(min(|LLR|,64)/64*65535)  [6] LLR sign errors  [7] reserved.
 * Which outputs are asked for selects the kernel instantiation: d_llr alone is the fastest
 * (HBM bound); d_hard alone is the reference's own chain ofdm::rx -> from_constl in one launch;
 * any other combination runs the all-outputs instantiation.  n_fft 64..8192 with even cp and
 * 16-byte aligned buffers take the tuned kernels, everything else the generic engines (same
 * results). */
#define SDR_DEMOD_WEIGHT_FROM_TAPS 1u
typedef struct sdr_demod_args {
    uint32_t struct_size;       /* = sizeof(sdr_demod_args); the round-2 layout (96 bytes, without `flags`) is accepted as flags = 0 */
    uint32_t n_fft;
    uint32_t cp;
    uint32_t mod;               /* sdr_modulation */
    uint64_t n_sym;
    uint32_t sym_per_set;
    float llr_weight0;
    const void* d_rx;           /* n_sym*(n_fft+cp) c32 */
    const void* d_eq;           /* n_sets*n_fft c32 or NULL */
    const float* d_llr_weight;  /* n_sets*n_fft or NULL */
    float* d_llr;
    uint8_t* d_hard;
    void* d_fft;
    const uint8_t* d_ref_hard;
    uint64_t* d_counters;
    uint32_t flags;             /* 0, or SDR_DEMOD_WEIGHT_FROM_TAPS */
    uint32_t reserved;          /* 0 */
} sdr_demod_args;
int sdr_ofdm_demod_c32(sdr_ctx* ctx, const sdr_demod_args* args);
/* Same with every pointer in HOST memory (copies in, runs, copies out, synchronises);
 * n_sets = number of tap / weight sets behind d_eq / d_llr_weight (ignored when both are NULL).
 * The kernel indexes set sym / sym_per_set, so n_sets must be at least
 * ceil(n_sym / sym_per_set) (1 when sym_per_set is 0): fewer is SDR_ERR_INVALID.  The device form
 * above cannot see the size of its arrays: there the same bound is the caller's contract. */
int sdr_ofdm_demod_host_c32(sdr_ctx* ctx, const sdr_demod_args* host_args, size_t n_sets);

/* sdr_ofdm_demod_c32 for callers that cannot allocate.  NO COUNTERPART IN THE REFERENCE: it
 * exists for code that must not call hipMalloc while it runs -- HIP graph capture, tight serving
 * loops, harnesses that forbid allocation.  Same arguments, same results, but the device memory
 * the kernels need beyond inputs and outputs (the twiddle table and the 1024 x 8 counter slots,
 * which sdr_ofdm_demod_c32 allocates lazily in the context) comes from one static device array of
 * 384 KiB per GPU (four counter-slot sets + the tables of n_fft = 64 ... 8192), resolved and filled
 * once per device by the first sdr_create on it; a failure of that set-up is reported HERE, not by
 * sdr_create.  The call enqueues one 64 KiB hipMemsetAsync (counting calls only), the receive
 * kernel and the counter fold -- no hipMalloc, hipFree or copy.  Limits: n_fft = 64 ... 8192, even
 * cp, 16-byte aligned buffers (what the tuned kernels take); anything else is
 * SDR_ERR_UNSUPPORTED.  Each live context owns one of the four slot sets of its GPU (claimed by
 * sdr_create, returned by sdr_destroy), so contexts do not race; a fifth live context on one GPU
 * has none and its COUNTING calls are refused with SDR_ERR_UNSUPPORTED (the others work).  As for
 * every entry point, one context must not be used from two threads at once. */
int sdr_ofdm_demod_noalloc_c32(sdr_ctx* ctx, const sdr_demod_args* args);

/* ---- Where the buffers lie in device memory (no counterpart in the reference: it has no device) ------------
 * On MI355X the rate of the receive kernel -- of any kernel that streams a large input into a larger output --
 * depends on WHERE in device memory the two buffers lie, in three steps measured as 0.74 / 0.77 / 0.80-0.83 of the
 * HBM peak for the SAME kernel on the SAME data (profiles/r3_placement_map.txt): a large
 * allocation is made of separately placed physical regions (32-64 GiB each on the boxes measured); the kernel is
 * slowest when input and output lie in one class of region, faster when they lie in different ones, and fastest
 * when the OUTPUT straddles two regions, so that the eight XCDs' write streams are spread over both.  Which it
 * gets from a plain allocation is chance (the 0.69-0.77 spread between processes of rounds 2-3).  A caller that
 * owns one large arena can choose: this call times a traffic model of sdr_ofdm_demod_c32 (the same reads and
 * writes in the same workgroup order, no arithmetic) for candidate placements of the two buffers inside
 * [d_arena, d_arena + arena_bytes) -- first the output at every multiple of step_bytes with the input out of its
 * way, then the input at every multiple of step_bytes beside the best output -- and returns the fastest pair of
 * byte offsets (multiples of 2 MiB).
 *   - The arena's contents are OVERWRITTEN where outputs were tried: fill the input afterwards.
 *   - The call enqueues on the context's stream and WAITS for its measurements (not for use under stream capture).
 *   - step_bytes = 0 means half the output buffer, at most 8 GiB and at least arena_bytes / 256.
 *   - n_fft 64 ... 8192, even cp, a known modulation, n_sym > 0, arena 16-byte aligned and large enough for both
 *     buffers; SDR_ERR_INVALID otherwise.  For ofdm::rx with bins out (sdr_ofdm_rx_c32) pass mod = SDR_MOD_QPSK:
 *     8 bytes per subcarrier either way.
 *   - *best_gbs / *worst_gbs (optional): the model's rate at the best and the worst candidate, in GB/s on
 *     SURVEY 8(d) bytes.
 *   - The arena must CONTAIN a boundary between two regions to gain anything: larger than the largest region
 *     (64 GiB on the boxes measured; 80 GiB was enough, 64 GiB was not).  best_gbs within a few percent of
 *     worst_gbs says that it does not.
 * About (2 * arena_bytes / step_bytes) * 3 launches of the model: a second for 1 Mi symbols of FFT-1024 in 200 GiB. */
int sdr_placement_search_c32(sdr_ctx* ctx, void* d_arena, uint64_t arena_bytes, uint32_t n_fft, uint32_t cp, uint32_t mod, uint64_t n_sym,
                             uint64_t step_bytes, uint64_t* in_offset, uint64_t* out_offset, float* best_gbs, float* worst_gbs);

/* Equaliser taps and LLR weights from a channel estimate H (n values):
 *   p = |H|^2, G = conj(H)/p, w = norm^2 * p * inv_n0. */
int sdr_zf_taps_c32(sdr_ctx* ctx, int mod, const void* d_chan, size_t n, float inv_n0, void* d_taps, float* d_weight);
int sdr_zf_taps_host_c32(sdr_ctx* ctx, int mod, const void* h_chan, size_t n, float inv_n0, void* h_taps, float* h_weight);

/* Soft demapping alone (X2) on equalised symbols already in memory. */
int sdr_soft_demap_c32(sdr_ctx* ctx, int mod, const void* d_sym, size_t n_sym, const float* d_weight, float weight0, float* d_llr);
int sdr_soft_demap_host_c32(sdr_ctx* ctx, int mod, const void* h_sym, size_t n_sym, const float* h_weight, float weight0, float* h_llr);

/* ---- X4: device frame generator, AWGN and Monte-Carlo link (no reference counterpart; model:
 * app/QFDemoWindow.cpp:201-224) --------------------------------------------------------------
 * Randomness is Philox4x32-10 keyed by `seed` and counted by (global symbol index, subcarrier /
 * sample index), so results do not depend on how symbols are split over calls or GPUs.
 * generate: symbols g0..g0+n_sym-1 -> to_constl -> [x H_k] -> tx (IFFT, CP) -> + N(0, sigma^2)
 * per real component.  d_chan: one set of n_fft values per sym_per_set symbols, first set =
 * g0/sym_per_set; NULL = flat.  d_tx_bits (optional) = transmitted bits, MSB-first stream. */
typedef struct sdr_link_args {
    uint32_t struct_size;
    uint32_t n_fft;
    uint32_t cp;
    uint32_t mod;
    uint64_t seed;
    uint64_t first_symbol;      /* g0 */
    uint64_t n_sym;
    uint32_t sym_per_set;
    float sigma;                /* per-component noise standard deviation, time domain */
    float llr_weight0;          /* ber only */
    uint32_t reserved;
    const void* d_chan;         /* generate only */
    void* d_rx;                 /* generate only: n_sym*(n_fft+cp) c32 */
    uint8_t* d_tx_bits;         /* generate only, optional */
    uint64_t* d_counters;       /* ber only: 8 x uint64, accumulated into */
} sdr_link_args;
int sdr_link_generate_c32(sdr_ctx* ctx, const sdr_link_args* args);
/* generate -> demodulate -> count entirely on the device; nothing but counters touches HBM. */
int sdr_link_ber_c32(sdr_ctx* ctx, const sdr_link_args* args);

/* A whole Eb/N0 sweep (BASELINE.json configs[4]) as ONE launch: `n_points` (modulation, noise
 * level) points, each with the symbol range this rank handles, generated -> demodulated ->
 * counted on the device; row i of d_counters (n_points x 8 uint64, accumulated into) receives
 * point i.  Same results as one sdr_link_ber_c32 call per point.  `points` is HOST memory (copied
 * to the device by the call, unless it is the table of the context's previous sweep, whose device copy
 * is reused); points with n_sym = 0 are skipped.  Model of the loop:
 * app/QFDemoWindow.cpp:201-224 (one symbol per tick there). */
/* SDR_POINT_HARD_ONLY: count bit and subcarrier-symbol errors from hard decisions alone, the way the
 * reference's own receive chain works (ofdm::rx -> from_constl, lib/inc/ofdm.hpp:37-43 +
 * modulation.hpp:96-112): no LLR is formed, counters 5 and 6 of the point are left as they were.
 * Counters 0-4 are identical to those of the full form. */
#define SDR_POINT_HARD_ONLY 1u
typedef struct sdr_sweep_point {
    uint32_t mod;               /* sdr_modulation */
    float sigma;                /* per-component noise standard deviation, time domain */
    float llr_weight0;
    uint32_t flags;             /* 0, or SDR_POINT_HARD_ONLY */
    uint64_t first_symbol;      /* global index of the first symbol of this point handled here */
    uint64_t n_sym;
} sdr_sweep_point;
/* SDR_SWEEP_ZERO_COUNTERS: the call clears the n_points x 8 counters on the context's stream ahead of the launch
 * (one hipMemsetAsync) instead of accumulating into what they hold: a sweep is then ONE library call. */
#define SDR_SWEEP_ZERO_COUNTERS 1u
typedef struct sdr_sweep_args {
    uint32_t struct_size;       /* = sizeof(sdr_sweep_args) */
    uint32_t n_fft;
    uint32_t cp;
    uint32_t n_points;
    uint64_t seed;
    const sdr_sweep_point* points;   /* host */
    uint64_t* d_counters;            /* device, n_points x 8 */
    uint32_t flags;                  /* 0, or SDR_SWEEP_ZERO_COUNTERS */
    uint32_t reserved;               /* 0 */
} sdr_sweep_args;
int sdr_link_sweep_c32(sdr_ctx* ctx, const sdr_sweep_args* args);

/* Channel realisations for the faded link: H[set][k], unit-variance complex Gaussian from Philox
 * block (k, set, stream tag); sets first_set .. first_set+n_sets-1, n_sets*n_fft c32 written. */
int sdr_link_channel_c32(sdr_ctx* ctx, uint32_t n_fft, uint64_t seed, uint64_t first_set, size_t n_sets, void* d_chan);

/* The whole faded link over MORE symbols than fit in memory (BASELINE.json configs[2]: FFT-2048,
 * 64-QAM, per-subcarrier equalisation, 4 Mi symbols = 86 GB of samples and 206 GB of LLRs): symbols
 * first_symbol .. first_symbol+n_sym-1 in resident chunks of chunk_symbols that REUSE one
 * workspace.  Per chunk: channel realisation per set of sym_per_set symbols -> zero-forcing taps and
 * LLR weights (sdr_zf_taps_c32, perfect channel knowledge) -> frame generator through that channel
 * + AWGN -> sdr_ofdm_demod_c32 with taps and weights (LLRs of the chunk into the workspace unless
 * no_llr, hard bits compared with the transmitted ones) -> d_counters (8 uint64, accumulated into).
 * The generator of chunk i+1 runs on a second stream beside the demodulation of chunk i (two sample
 * buffers).  Results do not depend on chunk_symbols.  first_symbol and chunk_symbols must be
 * multiples of sym_per_set (>= 1); n_fft 64..8192, even cp.  The workspace (device memory,
 * 256-byte aligned, sdr_link_chain_workspace bytes) is the caller's; d_llr_chunk (optional) receives
 * where the LLRs of the LAST chunk lie in it.  Model loop: app/QFDemoWindow.cpp:201-224. */
typedef struct sdr_chain_args {
    uint32_t struct_size;       /* = sizeof(sdr_chain_args) */
    uint32_t n_fft;
    uint32_t cp;
    uint32_t mod;               /* sdr_modulation */
    uint64_t seed;
    uint64_t first_symbol;
    uint64_t n_sym;
    uint64_t chunk_symbols;
    uint32_t sym_per_set;
    uint32_t no_llr;            /* 1: counters only, no LLR chunk in the workspace */
    float sigma;                /* per-component noise standard deviation, time domain */
    float inv_n0;               /* 1/N0 for the LLR weights */
    void* d_workspace;
    uint64_t workspace_bytes;
    uint64_t* d_counters;
    float** d_llr_chunk;        /* optional out (host pointer to a device pointer) */
} sdr_chain_args;
int sdr_link_chain_workspace(const sdr_chain_args* args, uint64_t* bytes);
int sdr_link_chain_c32(sdr_ctx* ctx, const sdr_chain_args* args);

/* ---- Streaming ingest ------------------------------------------------
 * What sits in front of the path in the reference is a host-side ring that is fed one OFDM
 * symbol at a time (utils::sliding_buffer, lib/inc/sliding_buffer.hpp:14-104, filled at
 * app/QFDemoWindow.cpp:213-229).  The device analogue is a ring of `depth` slots, each with a
 * pinned host input buffer, device buffers and pinned host result buffers.  The producer fills
 * a slot's input with up to chunk_symbols received symbols and submits it; upload, the fused
 * receive kernel (sdr_ofdm_demod_c32 on the slot) and the download of the results run on three
 * streams, so the upload of chunk i+1 and the download of chunk i-1 overlap the kernel of chunk i.
 *   acquire -> fill h_rx -> submit   (producer; acquire fails with SDR_ERR_INVALID when all
 *                                     slots are in flight: collect/release first)
 *   collect -> use results -> release (consumer; collect blocks until the oldest chunk is done)
 * Chunks complete in submission order.  One pipe must not be used from two threads at once,
 * except that one producer thread (acquire/submit) and one consumer thread (collect/release)
 * may run concurrently.  Equaliser sets restart at every chunk (set 0 = first symbol of the
 * chunk), so chunk_symbols should be a multiple of sym_per_set. */
typedef struct sdr_pipe sdr_pipe;
enum { SDR_PIPE_LLR_TO_HOST = 1, SDR_PIPE_HARD_TO_HOST = 2, SDR_PIPE_WANT_LLR = 4, SDR_PIPE_WANT_HARD = 8 };
typedef struct sdr_pipe_config {
    uint32_t struct_size;       /* = sizeof(sdr_pipe_config) */
    uint32_t n_fft;
    uint32_t cp;
    uint32_t mod;               /* sdr_modulation */
    uint64_t chunk_symbols;     /* capacity of one slot */
    uint32_t depth;             /* slots, 2..16 */
    uint32_t flags;             /* SDR_PIPE_*; *_TO_HOST implies the matching WANT */
    uint32_t sym_per_set;
    float llr_weight0;
    const void* d_eq;           /* device, n_sets*n_fft c32 or NULL; must outlive the pipe */
    const float* d_llr_weight;  /* device, n_sets*n_fft or NULL */
    /* The pipe's streams are non-blocking: whatever produced d_eq / d_llr_weight on the device (e.g.
     * sdr_zf_taps_c32 on another stream) must be COMPLETE before the first sdr_pipe_submit --
     * synchronise that stream first; nothing in the pipe orders its kernels after it. */
} sdr_pipe_config;
typedef struct sdr_pipe_result {
    uint64_t sequence;          /* 0,1,2,... in submission order */
    uint64_t n_sym;
    const float* h_llr;         /* pinned host, n_sym*n_fft*mod, or NULL */
    const uint8_t* h_hard;      /* pinned host, n_sym*n_fft*mod/8, or NULL */
    const float* d_llr;         /* device copies of the same, valid until release, or NULL */
    const uint8_t* d_hard;
} sdr_pipe_result;
int sdr_pipe_create(int device, const sdr_pipe_config* config, sdr_pipe** out);
int sdr_pipe_acquire(sdr_pipe* pipe, void** h_rx);
int sdr_pipe_submit(sdr_pipe* pipe, uint64_t n_sym);
int sdr_pipe_collect(sdr_pipe* pipe, sdr_pipe_result* result);
int sdr_pipe_release(sdr_pipe* pipe);
/* Chunks submitted and not yet released. */
int sdr_pipe_in_flight(sdr_pipe* pipe, uint32_t* count);
void sdr_pipe_destroy(sdr_pipe* pipe);

/* Host-visible pieces of the generator, for tests. */
int sdr_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
/* The tables of the Gaussian generator the kernels use: 32 pairs (r_i, L_i) of the logarithm's range reduction,
 * then 256 pairs (sin, cos)(2 pi j / 256); 576 floats written to host memory. */
int sdr_rng_tables_f32(float* h_out);

#ifdef __cplusplus
}
#endif
#endif /* SDR_AMD_H */
