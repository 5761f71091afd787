/* sdr_amd_multi.h -- C ABI of libsdr_amd_multi.so: the receive-side Monte-Carlo runs of libsdr_amd.so sharded
 * over several GPUs of one node, for callers written in C or C++ (the Python drivers bench.py / ber_sweep.py do the
 * same with one process per GPU and torch.distributed).
 *
 * NO COUNTERPART IN THE REFERENCE: the reference runs one OFDM symbol per timer tick on one CPU thread
 * (app/QFDemoWindow.cpp:201-224) and has no notion of a device, let alone of several.  OFDM symbols are independent
 * (ofdm::rx keeps no state, lib/inc/ofdm.hpp:37-43), so every GPU takes a contiguous range of whole frames of every
 * (modulation, Eb/N0) point -- random streams are keyed by the GLOBAL symbol index, so the summed counters do not
 * depend on the number of GPUs -- and the only exchange is ONE RCCL all-reduce (ncclSum over ncclUint64, a few KiB,
 * latency bound) of the counters per run, over xGMI.
 *
 * Two ways to form the communicator:
 *   sdr_multi_create       one host process drives all the GPUs (one context, one stream and one host thread per GPU;
 *                          ncclCommInitAll)
 *   sdr_multi_create_rank  one process per GPU (ncclCommInitRank); rank 0 makes the 128-byte id with
 *                          sdr_multi_unique_id and hands it to the other ranks by whatever means the job has
 * This library links RCCL; libsdr_amd.so itself does not.
 */
#ifndef SDR_AMD_MULTI_H
#define SDR_AMD_MULTI_H

#include "sdr_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct sdr_multi sdr_multi;

#define SDR_MULTI_UNIQUE_ID_BYTES 128

/* Message of the calling thread's last failure in this library (failures inside the per-GPU worker threads are
 * carried over to the caller's). */
const char* sdr_multi_last_error(void);

int sdr_multi_create(const int* devices, int n_devices, sdr_multi** out);
int sdr_multi_unique_id(void* id_out /* SDR_MULTI_UNIQUE_ID_BYTES */);
int sdr_multi_create_rank(int device, int rank, int world, const void* unique_id, sdr_multi** out);
void sdr_multi_destroy(sdr_multi* m);
/* GPUs this handle drives (1 for the one-process-per-GPU form) and ranks of the communicator. */
int sdr_multi_local_devices(const sdr_multi* m);
int sdr_multi_world(const sdr_multi* m);

/* BASELINE.json configs[4]: a whole Eb/N0 sweep.  args->points (host) describe the WHOLE job: point i covers symbols
 * first_symbol .. first_symbol + n_sym - 1, n_sym a multiple of frame_symbols.  Every rank takes a contiguous range of
 * whole frames of every point (rank r of W: frames [r*b + min(r,e), ...) with b, e = divmod(frames, W)), runs ONE
 * sdr_link_sweep_c32 launch on it, and one all-reduce sums the n_points x 8 counters, which are copied to h_counters
 * (host, n_points x 8 uint64, OVERWRITTEN; every rank receives the sums).  args->d_counters and args->flags are
 * ignored.  *seconds (optional) = wall time from the first launch to the end of the last synchronisation. */
int sdr_multi_sweep_c32(sdr_multi* m, const sdr_sweep_args* args, uint32_t frame_symbols, uint64_t* h_counters, double* seconds);

/* BASELINE.json configs[2] / [3]: the whole faded link (sdr_link_chain_c32: channel -> zero-forcing taps -> generator
 * -> equalised demodulation to LLRs -> counters, in resident chunks) with the symbol range split over the ranks in
 * whole sets of sym_per_set symbols; each GPU works in a workspace this handle allocates (and keeps), eight counters
 * are all-reduced into h_counters (host, OVERWRITTEN).  args->d_workspace, workspace_bytes, d_counters and
 * d_llr_chunk are ignored. */
int sdr_multi_chain_c32(sdr_multi* m, const sdr_chain_args* args, uint64_t* h_counters, double* seconds);

#ifdef __cplusplus
}
#endif
#endif /* SDR_AMD_MULTI_H */
