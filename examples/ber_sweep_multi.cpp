// ber_sweep_multi.cpp -- BASELINE.json configs[4] from C++: the Eb/N0 Monte-Carlo sweep (QPSK ... 256-QAM) sharded
// over the GPUs of one node, one RCCL all-reduce of the error counters (include/sdr_amd_multi.h).  The native twin of
// ber_sweep.py: same grid, same frame plan, same random streams, hence the same counters for any number of GPUs.
//
// The loop it scales up is the reference's only end-to-end one, app/QFDemoWindow.cpp:201-224 (bits -> to_constl ->
// ofdm::tx -> ofdm::rx -> from_constl, one symbol per tick, identity channel), with AWGN inserted and the
// demodulator's LLR statistics counted beside the bit errors.
//
//   ber_sweep_multi [--gpus N] [--bits B] [--n-fft N] [--cp C] [--mods 2 4 6 8] [--step-db S] [--seed K] [--hard-only]
//                   [--chain SYMBOLS]      (configs[2] instead: the faded FFT-2048 / 64-QAM link over SYMBOLS symbols)
//
// stderr: the BER table.  stdout: one JSON line.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "sdr_amd_multi.h"

namespace {

constexpr uint32_t kSymbolsPerFrame = 14;

double grid_energy(int bps)
{
    if (bps == 1) return 1.0;
    const double levels = static_cast<double>(1 << (bps / 2));
    return 2.0 * (levels * levels - 1.0) / 3.0;
}
// sdr_amd/link.py: noise_sigma, llr_weight, theory_ber ("AWGN scaling")
double noise_sigma(int n_fft, int bps, double ebn0_db)
{
    const double n0 = 1.0 / (bps * std::pow(10.0, ebn0_db / 10.0));
    return std::sqrt(n0 / (2.0 * n_fft));
}
double llr_weight(int bps, double ebn0_db)
{
    const double n0 = 1.0 / (bps * std::pow(10.0, ebn0_db / 10.0));
    return (1.0 / grid_energy(bps)) / n0;
}
double q_function(double x) { return 0.5 * std::erfc(x / std::sqrt(2.0)); }
double theory_ber(int bps, double ebn0_db)
{
    const double ebn0 = std::pow(10.0, ebn0_db / 10.0);
    if (bps <= 2) return q_function(std::sqrt(2.0 * ebn0));
    const double m = static_cast<double>(1 << bps);
    return (4.0 / bps) * (1.0 - 1.0 / std::sqrt(m)) * q_function(std::sqrt(3.0 * bps * ebn0 / (m - 1.0)));
}
int default_max_db(int bps) { return bps == 2 ? 10 : bps == 4 ? 14 : bps == 6 ? 18 : bps == 8 ? 24 : 12; }

struct Point { int bps; double ebn0_db; uint64_t frames, first_frame; };

} // namespace

int main(int argc, char** argv)
{
    int gpus = 0, n_fft = 1024, cp = 256;
    double bits = 1e10, step_db = 1.0;
    uint64_t seed = 5, chain_symbols = 0;
    bool hard_only = false;
    std::vector<int> mods{2, 4, 6, 8};
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        auto next = [&]() -> const char* { if (i + 1 >= argc) { std::fprintf(stderr, "%s needs a value\n", a.c_str()); std::exit(2); } return argv[++i]; };
        if (a == "--gpus") gpus = std::atoi(next());
        else if (a == "--bits") bits = std::atof(next());
        else if (a == "--n-fft") n_fft = std::atoi(next());
        else if (a == "--cp") cp = std::atoi(next());
        else if (a == "--step-db") step_db = std::atof(next());
        else if (a == "--seed") seed = std::strtoull(next(), nullptr, 10);
        else if (a == "--chain") chain_symbols = std::strtoull(next(), nullptr, 10);
        else if (a == "--hard-only") hard_only = true;
        else if (a == "--mods") { mods.clear(); while (i + 1 < argc && argv[i + 1][0] != '-') mods.push_back(std::atoi(argv[++i])); }
        else { std::fprintf(stderr, "unknown option %s\n", a.c_str()); return 2; }
    }
    int visible = 0;
    if (sdr_device_count(&visible) != SDR_OK || visible <= 0) { std::fprintf(stderr, "no GPU: %s\n", sdr_last_error()); return 1; }
    if (gpus <= 0 || gpus > visible) gpus = visible;
    std::vector<int> devices(static_cast<size_t>(gpus));
    for (int d = 0; d < gpus; ++d) devices[static_cast<size_t>(d)] = d;
    sdr_multi* m = nullptr;
    if (sdr_multi_create(devices.data(), gpus, &m) != SDR_OK) { std::fprintf(stderr, "%s\n", sdr_multi_last_error()); return 1; }

    if (chain_symbols) {
        // BASELINE.json configs[2]: FFT-2048, cp 512, 64-QAM, one channel realisation per 14-symbol frame, 18 dB
        sdr_chain_args c{};
        c.struct_size = sizeof(c);
        c.n_fft = 2048; c.cp = 512; c.mod = 6; c.seed = 3; c.first_symbol = 0; c.n_sym = chain_symbols;
        c.chunk_symbols = ((1u << 18) / kSymbolsPerFrame) * kSymbolsPerFrame; c.sym_per_set = kSymbolsPerFrame;
        c.sigma = static_cast<float>(noise_sigma(2048, 6, 18.0));
        c.inv_n0 = static_cast<float>(6 * std::pow(10.0, 1.8));
        uint64_t k[8] = {};
        double secs = 0.0;
        for (int pass = 0; pass < 2; ++pass)      // the first pass allocates the workspaces and builds the tables
            if (sdr_multi_chain_c32(m, &c, k, &secs) != SDR_OK) { std::fprintf(stderr, "%s\n", sdr_multi_last_error()); return 1; }
        std::printf("{\"config\": {\"workload\": \"fft2048_64qam_eq_chain\", \"symbols\": %llu}, \"n_gpus\": %d, \"seconds\": %.6f, \"symbols_per_s\": %.6g, "
                    "\"bit_errors\": %llu, \"bits\": %llu, \"ofdm_symbols\": %llu, \"collective\": \"one ncclAllReduce(ncclSum) of 8 ncclUint64\"}\n",
                    static_cast<unsigned long long>(chain_symbols), gpus, secs, chain_symbols / secs, static_cast<unsigned long long>(k[0]),
                    static_cast<unsigned long long>(k[1]), static_cast<unsigned long long>(k[4]));
        sdr_multi_destroy(m);
        return 0;
    }

    // the frame plan of sdr_amd/link.py: plan_sweep -- total bits split evenly over the points, whole 14-symbol frames,
    // every point its own range of global frame indices
    size_t n_points = 0;
    for (int bps : mods) n_points += static_cast<size_t>(std::lround(default_max_db(bps) / step_db)) + 1;
    const double bits_per_point = bits / static_cast<double>(n_points ? n_points : 1);
    std::vector<Point> pts;
    uint64_t next_frame = 0;
    for (int bps : mods) {
        const double frame_bits = static_cast<double>(kSymbolsPerFrame) * n_fft * bps;
        uint64_t frames = static_cast<uint64_t>(std::ceil(bits_per_point / frame_bits));
        if (frames < 1) frames = 1;
        const long count = std::lround(default_max_db(bps) / step_db) + 1;
        for (long e = 0; e < count; ++e) {
            pts.push_back({bps, e * step_db, frames, next_frame});
            next_frame += frames;
        }
    }
    std::vector<sdr_sweep_point> table(pts.size());
    for (size_t i = 0; i < pts.size(); ++i) {
        table[i].mod = static_cast<uint32_t>(pts[i].bps);
        table[i].sigma = static_cast<float>(noise_sigma(n_fft, pts[i].bps, pts[i].ebn0_db));
        table[i].llr_weight0 = static_cast<float>(llr_weight(pts[i].bps, pts[i].ebn0_db));
        table[i].flags = hard_only ? SDR_POINT_HARD_ONLY : 0u;
        table[i].first_symbol = pts[i].first_frame * kSymbolsPerFrame;
        table[i].n_sym = pts[i].frames * kSymbolsPerFrame;
    }
    sdr_sweep_args a{};
    a.struct_size = sizeof(a);
    a.n_fft = static_cast<uint32_t>(n_fft); a.cp = static_cast<uint32_t>(cp); a.n_points = static_cast<uint32_t>(table.size());
    a.seed = seed; a.points = table.data();
    std::vector<uint64_t> counters(table.size() * 8);
    double secs = 0.0;
    for (int pass = 0; pass < 2; ++pass)      // the first pass builds the twiddle / generator tables
        if (sdr_multi_sweep_c32(m, &a, kSymbolsPerFrame, counters.data(), &secs) != SDR_OK) { std::fprintf(stderr, "%s\n", sdr_multi_last_error()); return 1; }

    unsigned long long total_bits = 0, total_errors = 0, total_frames = 0;
    std::fprintf(stderr, "%4s %6s %12s %10s %11s %11s\n", "mod", "Eb/N0", "bits", "errors", "BER", "theory");
    for (size_t i = 0; i < pts.size(); ++i) {
        const uint64_t* r = &counters[i * 8];
        std::fprintf(stderr, "%4d %6.1f %12llu %10llu %11.3e %11.3e\n", pts[i].bps, pts[i].ebn0_db, static_cast<unsigned long long>(r[1]),
                     static_cast<unsigned long long>(r[0]), r[1] ? static_cast<double>(r[0]) / static_cast<double>(r[1]) : 0.0, theory_ber(pts[i].bps, pts[i].ebn0_db));
        total_bits += r[1]; total_errors += r[0]; total_frames += pts[i].frames;
    }
    std::printf("{\"metric\": \"Monte-Carlo BER sweep throughput (generate + AWGN + demodulate + count on device)\", \"value\": %.6g, \"unit\": \"frames/s\", "
                "\"bits_per_s\": %.6g, \"n_gpus\": %d, \"seconds_per_sweep\": %.6f, \"points\": %zu, \"total_bits\": %llu, \"checksum_bit_errors\": %llu, "
                "\"collective\": \"one ncclAllReduce(ncclSum) of %zu ncclUint64\", \"host\": \"C++ (libsdr_amd_multi.so)\", \"counters\": [",
                total_frames / secs, total_bits / secs, gpus, secs, pts.size(), total_bits, total_errors, counters.size());
    for (size_t i = 0; i < pts.size(); ++i) {
        std::printf("%s[", i ? ", " : "");
        for (int j = 0; j < 8; ++j) std::printf("%s%llu", j ? ", " : "", static_cast<unsigned long long>(counters[i * 8 + static_cast<size_t>(j)]));
        std::printf("]");
    }
    std::printf("]}\n");
    sdr_multi_destroy(m);
    return 0;
}
