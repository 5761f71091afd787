// multi.hip -- libsdr_amd_multi.so: the Monte-Carlo runs of libsdr_amd.so sharded over the GPUs of one node with ONE
// RCCL all-reduce of the counters per run (include/sdr_amd_multi.h).  Host code only: every kernel is libsdr_amd's.
//
// No reference counterpart (the reference's loop, app/QFDemoWindow.cpp:201-224, is one symbol per tick on one CPU
// thread).  What is exchanged is n_points x 8 (or 8) unsigned 64-bit integers: latency bound, xGMI bandwidth is
// irrelevant, and integer sums make the result independent of the reduction order and of the number of GPUs.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <chrono>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "sdr_amd_multi.h"

namespace {

thread_local std::string g_error;

int fail(int code, const char* fmt, ...)
{
    char buf[640];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_error = buf;
    return code;
}

// one GPU of the handle
struct Lane {
    int device = 0;
    int rank = 0;                        // rank in the communicator
    sdr_ctx* ctx = nullptr;
    ncclComm_t comm = nullptr;
    uint64_t* d_counters = nullptr;      // grown on demand
    size_t counter_cap = 0;              // in uint64
    void* workspace = nullptr;           // sdr_link_chain_c32's, grown on demand
    size_t workspace_cap = 0;
    int status = SDR_OK;                 // of the last run, with its text
    std::string error;
};

int lane_fail(Lane& l, int code, const std::string& text)
{
    l.status = code;
    l.error = "GPU " + std::to_string(l.device) + ": " + text;
    return code;
}

#define LANE_HIP(l, expr)                                                                                   \
    do {                                                                                                    \
        hipError_t e_ = (expr);                                                                             \
        if (e_ != hipSuccess) return lane_fail(l, SDR_ERR_HIP, std::string("HIP error: ") + hipGetErrorString(e_) + " in " #expr); \
    } while (0)
#define LANE_NCCL(l, expr)                                                                                  \
    do {                                                                                                    \
        ncclResult_t r_ = (expr);                                                                           \
        if (r_ != ncclSuccess) return lane_fail(l, SDR_ERR_HIP, std::string("RCCL error: ") + ncclGetErrorString(r_) + " in " #expr); \
    } while (0)
#define LANE_SDR(l, expr)                                                                                   \
    do {                                                                                                    \
        int rc_ = (expr);                                                                                   \
        if (rc_ != SDR_OK) return lane_fail(l, rc_, sdr_last_error());                                      \
    } while (0)

int ensure_counters(Lane& l, size_t n_u64)
{
    if (l.counter_cap >= n_u64) return SDR_OK;
    if (l.d_counters) { LANE_HIP(l, hipFree(l.d_counters)); l.d_counters = nullptr; l.counter_cap = 0; }
    LANE_HIP(l, hipMalloc(reinterpret_cast<void**>(&l.d_counters), n_u64 * sizeof(uint64_t)));
    l.counter_cap = n_u64;
    return SDR_OK;
}

// contiguous share of `units` for rank r of w (sdr_amd/link.py: shard_frames)
void shard(uint64_t units, int r, int w, uint64_t& first, uint64_t& count)
{
    const uint64_t base = units / static_cast<uint64_t>(w), extra = units % static_cast<uint64_t>(w);
    const uint64_t ur = static_cast<uint64_t>(r);
    first = ur * base + (ur < extra ? ur : extra);
    count = base + (ur < extra ? 1 : 0);
}

} // namespace

struct sdr_multi {
    std::vector<Lane> lanes;
    int world = 0;
};

namespace {

// runs fn(lane) on one host thread per GPU (directly when there is one) and carries the first failure to the caller
template <typename Fn>
int run_lanes(sdr_multi* m, Fn&& fn)
{
    for (Lane& l : m->lanes) { l.status = SDR_OK; l.error.clear(); }
    if (m->lanes.size() == 1) {
        fn(m->lanes[0]);
    } else {
        std::vector<std::thread> pool;
        pool.reserve(m->lanes.size());
        for (Lane& l : m->lanes) pool.emplace_back([&fn, &l] { fn(l); });
        for (std::thread& t : pool) t.join();
    }
    for (Lane& l : m->lanes)
        if (l.status != SDR_OK) return fail(l.status, "%s", l.error.c_str());
    return SDR_OK;
}

int all_reduce_and_fetch(Lane& l, size_t n_u64, uint64_t* h_out)
{
    hipStream_t stream = static_cast<hipStream_t>(sdr_stream(l.ctx));
    // the only collective of a run: a few KiB of integers over xGMI, enqueued behind the kernels on the context's stream
    LANE_NCCL(l, ncclAllReduce(l.d_counters, l.d_counters, n_u64, ncclUint64, ncclSum, l.comm, stream));
    if (h_out) LANE_HIP(l, hipMemcpyAsync(h_out, l.d_counters, n_u64 * sizeof(uint64_t), hipMemcpyDeviceToHost, stream));
    LANE_HIP(l, hipStreamSynchronize(stream));
    return SDR_OK;
}

void destroy_lanes(sdr_multi* m)
{
    for (Lane& l : m->lanes) {
        (void)hipSetDevice(l.device);
        if (l.comm) (void)ncclCommDestroy(l.comm);
        if (l.d_counters) (void)hipFree(l.d_counters);
        if (l.workspace) (void)hipFree(l.workspace);
        if (l.ctx) sdr_destroy(l.ctx);
    }
    m->lanes.clear();
}

} // namespace

extern "C" {

const char* sdr_multi_last_error(void) { return g_error.c_str(); }

int sdr_multi_create(const int* devices, int n_devices, sdr_multi** out)
{
    if (!out) return fail(SDR_ERR_INVALID, "null out pointer");
    *out = nullptr;
    if (!devices || n_devices <= 0) return fail(SDR_ERR_INVALID, "no devices given");
    for (int i = 0; i < n_devices; ++i)
        for (int j = 0; j < i; ++j)
            if (devices[i] == devices[j]) return fail(SDR_ERR_INVALID, "device %d listed twice (RCCL wants one rank per GPU)", devices[i]);
    sdr_multi* m = new sdr_multi();
    m->world = n_devices;
    m->lanes.resize(static_cast<size_t>(n_devices));
    for (int i = 0; i < n_devices; ++i) {
        Lane& l = m->lanes[static_cast<size_t>(i)];
        l.device = devices[i];
        l.rank = i;
        if (sdr_create(devices[i], &l.ctx) != SDR_OK) {
            const int rc = fail(SDR_ERR_HIP, "GPU %d: %s", devices[i], sdr_last_error());
            destroy_lanes(m);
            delete m;
            return rc;
        }
    }
    std::vector<ncclComm_t> comms(static_cast<size_t>(n_devices), nullptr);
    const ncclResult_t r = ncclCommInitAll(comms.data(), n_devices, devices);
    if (r != ncclSuccess) {
        const int rc = fail(SDR_ERR_HIP, "RCCL error: %s in ncclCommInitAll over %d GPUs", ncclGetErrorString(r), n_devices);
        destroy_lanes(m);
        delete m;
        return rc;
    }
    for (int i = 0; i < n_devices; ++i) m->lanes[static_cast<size_t>(i)].comm = comms[static_cast<size_t>(i)];
    *out = m;
    return SDR_OK;
}

int sdr_multi_unique_id(void* id_out)
{
    static_assert(sizeof(ncclUniqueId) <= SDR_MULTI_UNIQUE_ID_BYTES, "the id fits the buffer the header promises");
    if (!id_out) return fail(SDR_ERR_INVALID, "null id buffer");
    ncclUniqueId id;
    const ncclResult_t r = ncclGetUniqueId(&id);
    if (r != ncclSuccess) return fail(SDR_ERR_HIP, "RCCL error: %s in ncclGetUniqueId", ncclGetErrorString(r));
    memset(id_out, 0, SDR_MULTI_UNIQUE_ID_BYTES);
    memcpy(id_out, &id, sizeof(id));
    return SDR_OK;
}

int sdr_multi_create_rank(int device, int rank, int world, const void* unique_id, sdr_multi** out)
{
    if (!out) return fail(SDR_ERR_INVALID, "null out pointer");
    *out = nullptr;
    if (!unique_id || world <= 0 || rank < 0 || rank >= world) return fail(SDR_ERR_INVALID, "bad rank %d / world %d / id", rank, world);
    sdr_multi* m = new sdr_multi();
    m->world = world;
    m->lanes.resize(1);
    Lane& l = m->lanes[0];
    l.device = device;
    l.rank = rank;
    if (sdr_create(device, &l.ctx) != SDR_OK) {
        const int rc = fail(SDR_ERR_HIP, "GPU %d: %s", device, sdr_last_error());
        destroy_lanes(m);
        delete m;
        return rc;
    }
    ncclUniqueId id;
    memcpy(&id, unique_id, sizeof(id));
    const ncclResult_t r = ncclCommInitRank(&l.comm, world, id, rank);      // sdr_create left `device` current
    if (r != ncclSuccess) {
        const int rc = fail(SDR_ERR_HIP, "RCCL error: %s in ncclCommInitRank (rank %d of %d)", ncclGetErrorString(r), rank, world);
        destroy_lanes(m);
        delete m;
        return rc;
    }
    *out = m;
    return SDR_OK;
}

void sdr_multi_destroy(sdr_multi* m)
{
    if (!m) return;
    destroy_lanes(m);
    delete m;
}

int sdr_multi_local_devices(const sdr_multi* m) { return m ? static_cast<int>(m->lanes.size()) : 0; }
int sdr_multi_world(const sdr_multi* m) { return m ? m->world : 0; }

int sdr_multi_sweep_c32(sdr_multi* m, const sdr_sweep_args* args, uint32_t frame_symbols, uint64_t* h_counters, double* seconds)
{
    if (!m || !args || !h_counters) return fail(SDR_ERR_INVALID, "null argument");
    if (args->struct_size != sizeof(sdr_sweep_args)) return fail(SDR_ERR_INVALID, "sdr_sweep_args.struct_size %u != %zu", args->struct_size, sizeof(sdr_sweep_args));
    if (frame_symbols == 0) return fail(SDR_ERR_INVALID, "frame_symbols must be >= 1");
    if (args->n_points && !args->points) return fail(SDR_ERR_INVALID, "null points");
    for (uint32_t i = 0; i < args->n_points; ++i)
        if (args->points[i].n_sym % frame_symbols != 0)
            return fail(SDR_ERR_INVALID, "point %u: %llu symbols are not whole frames of %u", i, static_cast<unsigned long long>(args->points[i].n_sym), frame_symbols);
    const size_t n_u64 = static_cast<size_t>(args->n_points) * 8;
    if (n_u64 == 0) { if (seconds) *seconds = 0.0; return SDR_OK; }
    const auto t0 = std::chrono::steady_clock::now();
    const int world = m->world;
    const int rc = run_lanes(m, [&](Lane& l) -> int {
        LANE_HIP(l, hipSetDevice(l.device));
        if (ensure_counters(l, n_u64) != SDR_OK) return l.status;
        // this rank's share of every point; a rank without frames of a point keeps the row with 0 symbols, so that
        // the rows line up across ranks
        std::vector<sdr_sweep_point> mine(args->points, args->points + args->n_points);
        for (sdr_sweep_point& p : mine) {
            uint64_t first = 0, count = 0;
            shard(p.n_sym / frame_symbols, l.rank, world, first, count);
            p.first_symbol += first * frame_symbols;
            p.n_sym = count * frame_symbols;
        }
        sdr_sweep_args a = *args;
        a.points = mine.data();
        a.d_counters = l.d_counters;
        a.flags = SDR_SWEEP_ZERO_COUNTERS;
        a.reserved = 0;
        LANE_SDR(l, sdr_link_sweep_c32(l.ctx, &a));          // copies the table before it returns
        return all_reduce_and_fetch(l, n_u64, l.rank == m->lanes[0].rank ? h_counters : nullptr);
    });
    if (seconds) *seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return rc;
}

int sdr_multi_chain_c32(sdr_multi* m, const sdr_chain_args* args, uint64_t* h_counters, double* seconds)
{
    if (!m || !args || !h_counters) return fail(SDR_ERR_INVALID, "null argument");
    if (args->struct_size != sizeof(sdr_chain_args)) return fail(SDR_ERR_INVALID, "sdr_chain_args.struct_size %u != %zu", args->struct_size, sizeof(sdr_chain_args));
    if (args->sym_per_set == 0) return fail(SDR_ERR_INVALID, "sym_per_set must be >= 1");
    const auto t0 = std::chrono::steady_clock::now();
    const int world = m->world;
    const uint64_t sets = (args->n_sym + args->sym_per_set - 1) / args->sym_per_set;
    const int rc = run_lanes(m, [&](Lane& l) -> int {
        LANE_HIP(l, hipSetDevice(l.device));
        if (ensure_counters(l, 8) != SDR_OK) return l.status;
        hipStream_t stream = static_cast<hipStream_t>(sdr_stream(l.ctx));
        LANE_HIP(l, hipMemsetAsync(l.d_counters, 0, 8 * sizeof(uint64_t), stream));
        uint64_t first_set = 0, n_sets = 0;
        shard(sets, l.rank, world, first_set, n_sets);
        sdr_chain_args a = *args;
        const uint64_t begin = first_set * args->sym_per_set;
        const uint64_t end = (first_set + n_sets) * args->sym_per_set < args->n_sym ? (first_set + n_sets) * args->sym_per_set : args->n_sym;
        a.first_symbol = args->first_symbol + begin;
        a.n_sym = end > begin ? end - begin : 0;
        a.d_llr_chunk = nullptr;
        if (a.n_sym) {
            uint64_t need = 0;
            LANE_SDR(l, sdr_link_chain_workspace(&a, &need));
            if (l.workspace_cap < need) {
                if (l.workspace) { LANE_HIP(l, hipStreamSynchronize(stream)); LANE_HIP(l, hipFree(l.workspace)); l.workspace = nullptr; l.workspace_cap = 0; }
                LANE_HIP(l, hipMalloc(&l.workspace, need));
                l.workspace_cap = need;
            }
            a.d_workspace = l.workspace;
            a.workspace_bytes = l.workspace_cap;
            a.d_counters = l.d_counters;
            LANE_SDR(l, sdr_link_chain_c32(l.ctx, &a));
        }
        return all_reduce_and_fetch(l, 8, l.rank == m->lanes[0].rank ? h_counters : nullptr);
    });
    if (seconds) *seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return rc;
}

} // extern "C"
