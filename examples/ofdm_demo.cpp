// ofdm_demo.cpp -- the reference's demo loop without the window.
//
// app/QFDemoWindow.cpp:201-266 does, on every timer tick: take the next 4 payload bytes ->
// modulation::to_constl<e16QAM> -> ofdm::tx(cp = 8) -> ofdm::rx -> modulation::from_constl ->
// append to a 50-character running text; it also keeps the last 512 transmitted samples and the
// tick's constellation points for its plots.  This program runs the same chain through the
// drop-in headers (every transform on the MI355X) and prints what the window would show:
//
//   ofdm_demo [--ticks N] [--time-csv FILE] [--constellation-csv FILE]
//
// stdout: the running text after the last tick, one line.  Exit status 1 if the decoded stream
// ever differs from the payload.
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "sdrlib/modulation.hpp"
#include "sdrlib/ofdm.hpp"
#include "sdrlib/sliding_buffer.hpp"

int main(int argc, char** argv)
{
    long ticks = 100;
    const char* time_csv = nullptr;
    const char* constl_csv = nullptr;
    for (int i = 1; i < argc; ++i) {
        if (!std::strcmp(argv[i], "--ticks") && i + 1 < argc) ticks = std::atol(argv[++i]);
        else if (!std::strcmp(argv[i], "--time-csv") && i + 1 < argc) time_csv = argv[++i];
        else if (!std::strcmp(argv[i], "--constellation-csv") && i + 1 < argc) constl_csv = argv[++i];
        else { std::fprintf(stderr, "usage: %s [--ticks N] [--time-csv FILE] [--constellation-csv FILE]\n", argv[0]); return 2; }
    }

    // the demo's payload and windows (QFDemoWindow.cpp:20-28)
    const std::string payload =
        "Hello, world! "
        "I am a Software-Defined Radio Stack.          "
        "This string is a result of demultiplexing a 16-QAM "
        "multiplexed OFDM signal.          ";
    utils::sliding_buffer<std::complex<float>> plot(512);
    utils::sliding_buffer<uint8_t> text(50);
    std::size_t pos = 0;
    std::vector<std::complex<double>> last_points;
    bool intact = true;

    for (long tick = 0; tick < ticks; ++tick) {
        std::vector<uint8_t> input;
        for (int i = 0; i < 4; ++i) input.push_back(static_cast<uint8_t>(payload[pos++ % payload.size()]));

        auto points = modulation::to_constl<modulation::e16QAM>(input);
        if (points.empty()) continue;
        auto tx = ofdm::tx(points, 8);
        if (!tx) { std::fprintf(stderr, "tx: %s\n", tx.error().c_str()); return 1; }
        auto rx = ofdm::rx(*tx, 8);
        if (!rx) { std::fprintf(stderr, "rx: %s\n", rx.error().c_str()); return 1; }
        auto bytes = modulation::from_constl<modulation::e16QAM>(*rx);
        if (bytes != input) intact = false;
        text.push_back(bytes.begin(), bytes.end());

        std::vector<std::complex<float>> narrow(tx->begin(), tx->end());
        plot.push_back(narrow.begin(), narrow.end());
        last_points = points;
    }

    if (time_csv) {
        if (FILE* f = std::fopen(time_csv, "w")) {
            std::fprintf(f, "n,re,im\n");
            for (std::size_t i = 0; i < plot.size(); ++i) std::fprintf(f, "%zu,%.9g,%.9g\n", i, plot[i].real(), plot[i].imag());
            std::fclose(f);
        }
    }
    if (constl_csv) {
        if (FILE* f = std::fopen(constl_csv, "w")) {
            std::fprintf(f, "i,q\n");
            for (const auto& p : last_points) std::fprintf(f, "%.17g,%.17g\n", p.real(), p.imag());
            std::fclose(f);
        }
    }
    std::string shown(text.begin(), text.end());
    for (char& c : shown) if (c == '\0') c = ' ';
    std::printf("%s\n", shown.c_str());
    return intact ? 0 : 1;
}
