// dropin_test.cpp -- the reference's gtest cases (lib/tests/fft_test.cpp, lib/tests/ofdm_test.cpp)
// restated against the drop-in headers in include/sdrlib/, with a self-contained harness (gtest
// is not installed in this image).  Needs a GPU; with --expect-no-device it instead checks that
// the absence of one is reported as an error value (no CPU fallback, no abort).
#include <cmath>
#include <complex>
#include <cstdio>
#include <cstring>
#include <deque>
#include <string>
#include <utility>
#include <vector>

#include "sdrlib/fft.hpp"
#include "sdrlib/modulation.hpp"
#include "sdrlib/ofdm.hpp"
#include "sdrlib/pipe.hpp"

// The CPU oracle (oracle/sdr_oracle.cpp, test infrastructure) as the arbiter of the equalised path: the same C entry
// points tests/ reach through ctypes.
extern "C" {
int orc_link_channel_c32(std::uint32_t n_fft, std::uint64_t seed, std::uint64_t first_set, std::size_t n_sets, float* chan_out);
int orc_zf_taps_f32(int bps, const float* chan, std::size_t n, float inv_n0, float* taps, float* wgt);
int orc_link_generate_c32(std::uint32_t n_fft, std::uint32_t cp, int bps, std::uint64_t seed, std::uint64_t g0, std::size_t n_sym, std::uint32_t sym_per_set,
                          float sigma, const float* chan, float* rx_out, std::uint8_t* tx_bits);
int orc_demod_c32(std::uint32_t n_fft, std::uint32_t cp, int bps, std::size_t n_sym, std::uint32_t sym_per_set, const float* rx, const float* eq, const float* wgt,
                  float w0, float* fft_out, float* llr_out, std::uint8_t* hard_out);
}

static int g_failed = 0;
#define CHECK(cond) do { if (!(cond)) { std::printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #cond); ++g_failed; } } while (0)

using namespace std::complex_literals;

static void fft_roundtrip_double()   // FFTTest.TransformsSequenceForthAndBackCorrectly / FftIfftForDouble (fft_test.cpp:10-38)
{
    std::vector<std::complex<double>> ref{0, 1, 2, 3, 4, 5, 6, 7}, seq = ref;
    CHECK(fft::fft2(seq.begin(), seq.end()).has_value());
    CHECK(std::abs(seq[0] - 28.0) < 1e-9);
    CHECK(fft::ifft2(seq.begin(), seq.end()).has_value());
    for (size_t i = 0; i < ref.size(); ++i) CHECK(std::abs(seq[i] - ref[i]) < 1e-9);
}

static void fft_roundtrip_float()    // FFTTest.FftIfftForFloat (fft_test.cpp:40-65)
{
    std::vector<std::complex<float>> ref{0, 1, 2, 3, 4, 5, 6, 7}, seq = ref;
    CHECK(fft::fft2(seq.begin(), seq.end()).has_value());
    CHECK(seq[1].real() == -3.99999952f);          // the reference's own rounding
    CHECK(fft::ifft2(seq.begin(), seq.end()).has_value());
    auto close = [](float x, float y) { float d = std::abs(x - y), m = std::max(std::abs(x), std::abs(y)); return d <= 1e-5f || d <= 1e-6f * m; };
    for (size_t i = 0; i < ref.size(); ++i) CHECK(close(seq[i].real(), ref[i].real()) && close(seq[i].imag(), ref[i].imag()));
}

static void fft_non_pow2_fails()     // FFTTest.SizeNonMultipleOf2Fails (fft_test.cpp:67-72)
{
    std::vector<std::complex<double>> seq{0, 1, 2};
    auto r = fft::fft2(seq.begin(), seq.end());
    CHECK(!r.has_value());
    CHECK(r.error() == "The sequence size must be of powers of 2");
    CHECK(seq[1] == 1.0);
}

static void ofdm_roundtrip()         // OFDMTest.TransformsSequenceForthAndBackCorrectly (ofdm_test.cpp:11-26)
{
    std::vector<std::complex<double>> ref{0.0 + 0i, 1.0 + 1i, 2.0 + 2i, -1.0 - 1i, -2.0 - 2i, -3.0 - 3i, 1.0 - 1i, -1.0 + 1i};
    std::vector<std::complex<double>> buf, res;
    CHECK(ofdm::tx(ref, 8, buf).has_value());
    CHECK(buf.size() == 16);
    CHECK(ofdm::rx(buf, 8, res).has_value());
    CHECK(res.size() == 8);
    for (size_t i = 0; i < ref.size(); ++i) CHECK(std::abs(res[i] - ref[i]) < 1e-9);
}

static void qam16_hello()            // OFDMTest.Transforms16QAMForthAndBackCorrectly (ofdm_test.cpp:28-36)
{
    std::vector<uint8_t> in{'H', 'e', 'l', 'l', 'o'};
    auto buf = modulation::to_constl<modulation::e16QAM>(in);          // T = double by default, as in the reference
    CHECK(buf.size() == 10);
    auto res = modulation::from_constl<modulation::e16QAM>(buf);
    CHECK(res == in);
}

static void app_tick()               // the GUI loop's call chain, app/QFDemoWindow.cpp:210-221
{
    std::vector<uint8_t> input{'S', 'D', 'R', '!'};
    auto syms = modulation::to_constl<modulation::e16QAM>(input);
    auto tx = ofdm::tx(syms, 8);
    CHECK(tx.has_value());
    std::vector<uint8_t> got;
    ofdm::rx(*tx, 8).transform([&](auto&& rx) { got = modulation::from_constl<modulation::e16QAM>(rx); });
    CHECK(got == input);
}

static void extras()
{
    // value-returning overloads, sizes 0/1, error paths, non-contiguous iterators
    std::vector<std::complex<float>> one{{3.f, 4.f}};
    CHECK(fft::fft2(one.begin(), one.end()).has_value() && one[0] == std::complex<float>(3.f, 4.f));   // fft.hpp:91
    std::vector<std::complex<float>> none;
    CHECK(fft::fft2(none.begin(), none.end()).has_value());
    std::deque<std::complex<double>> dq{0, 1, 2, 3, 4, 5, 6, 7};
    std::vector<std::complex<double>> v(dq.begin(), dq.end());
    CHECK(fft::fft2(dq.begin(), dq.end()).has_value());
    CHECK(fft::fft2(v.begin(), v.end()).has_value());
    for (size_t i = 0; i < v.size(); ++i) CHECK(dq[i] == v[i]);
    std::vector<std::complex<double>> six(6), out;
    CHECK(!ofdm::rx(six, 0, out).has_value());                           // 6 is not a power of two
    CHECK(out.size() == 6);
    CHECK(!ofdm::rx(six, 9, out).has_value());                           // shorter than its prefix
    auto r = ofdm::rx(std::vector<std::complex<float>>(80), 16);
    CHECK(r.has_value() && r->size() == 64);
    // e16QAM members (modulation.hpp:17-50)
    static_assert(modulation::e16QAM::bits_per_symbol == 4);
    CHECK(modulation::e16QAM::norm<float> == 1.0f / std::sqrt(10.0f));
    CHECK(modulation::e16QAM::inorm<float> == std::sqrt(10.0f));
    CHECK(modulation::e16QAM::norm<double> == 1.0 / std::sqrt(10.0));
    CHECK(modulation::e16QAM::inorm<double> == std::sqrt(10.0));
    CHECK(modulation::e16QAM::table<double>[0] == std::complex<double>(-3, -3));
    CHECK(modulation::e16QAM::table<double>[6] == std::complex<double>(-1, 3));
    CHECK(modulation::e16QAM::table<float>[15] == std::complex<float>(1, 1));
    static_assert(noexcept(modulation::e16QAM::nearest(std::declval<const std::complex<double>&>())) && noexcept(modulation::e16QAM::nearest(std::declval<const std::complex<float>&>())),
                  "nearest is noexcept like the reference's (modulation.hpp:50)");
    CHECK(modulation::e16QAM::nearest(std::complex<double>(0.9, -0.2)) == 9);   // +3, -1
    CHECK(modulation::e16QAM::nearest(std::complex<float>(0.f, 0.f)) == 5);     // four-way tie: lowest index
    // odd symbol count: trailing symbol dropped (modulation.hpp:103)
    auto s3 = modulation::to_constl<modulation::e16QAM, float>({0xAB, 0xCD});
    s3.pop_back();
    CHECK(modulation::from_constl<modulation::e16QAM>(s3) == std::vector<uint8_t>{0xAB});
    // other orders round trip, batched API, soft outputs
    std::vector<uint8_t> bytes{1, 2, 3, 250, 128, 77};
    CHECK(modulation::from_constl<modulation::eQPSK>(modulation::to_constl<modulation::eQPSK, float>(bytes)) == bytes);
    CHECK(modulation::from_constl<modulation::e64QAM>(modulation::to_constl<modulation::e64QAM, float>(bytes)) == bytes);
    CHECK(modulation::from_constl<modulation::e256QAM>(modulation::to_constl<modulation::e256QAM>(bytes)) == bytes);
    std::vector<uint8_t> payload(64 * 4 * 3 / 8);      // 96 bytes = 192 symbols = 3 OFDM symbols of 64 bins
    for (size_t i = 0; i < payload.size(); ++i) payload[i] = uint8_t(i * 37 + 11);
    auto freq = modulation::to_constl<modulation::e16QAM, float>(payload);
    std::vector<std::complex<float>> air, bins;
    CHECK(ofdm::tx_batch(freq, 64, 16, air).has_value() && air.size() == 3 * 80);
    CHECK(ofdm::rx_batch(air, 64, 16, bins).has_value() && bins.size() == 192);
    CHECK(modulation::from_constl<modulation::e16QAM>(bins) == payload);
    auto d = ofdm::demod_batch<modulation::e16QAM>(air, 64, 16, 2.5f);
    CHECK(d.has_value() && d->hard == payload && d->llr.size() == 192 * 4);
    auto l = modulation::llr<modulation::e16QAM>(bins, 2.5f);
    CHECK(l.size() == d->llr.size());
    for (size_t i = 0; i < l.size(); ++i) CHECK(l[i] == d->llr[i]);
    for (size_t k = 0; k < 192; ++k)
        for (int b = 0; b < 4; ++b) {
            const int bit = (payload[k / 2] >> ((k % 2 ? 0 : 4) + 3 - b)) & 1;
            CHECK((l[k * 4 + b] < 0) == (bit == 1));
        }
}

// FFT -> equalise -> LLR behind the drop-in headers (the step north_star inserts between ofdm::rx, lib/inc/ofdm.hpp:45-54, and
// modulation::from_constl, lib/inc/modulation.hpp:96-112): ofdm::zf_equaliser + ofdm::demod_batch(in, n, cp, equaliser)
// against the oracle on the same samples, bit for bit; the same cases tests/test_gpu_parity.py pins through Python.
template <typename Mod>
static void equalised_demod(std::uint32_t n, std::uint32_t cp, std::size_t n_sym, std::uint32_t per, float sigma)
{
    constexpr int bps = int(Mod::bits_per_symbol);
    const std::size_t n_sets = (n_sym + per - 1) / per;
    std::vector<std::complex<float>> chan(n_sets * n);
    CHECK(orc_link_channel_c32(n, 41, 0, n_sets, reinterpret_cast<float*>(chan.data())) == 0);
    for (auto& h : chan) if (std::abs(h) < 0.1f) h += 0.3f;                    // keep |H| away from zero
    std::vector<std::complex<float>> air(n_sym * (n + cp));
    std::vector<std::uint8_t> bits(n_sym * n * bps / 8);
    CHECK(orc_link_generate_c32(n, cp, bps, 9, 0, n_sym, per, sigma, reinterpret_cast<const float*>(chan.data()), reinterpret_cast<float*>(air.data()), bits.data()) == 0);
    std::vector<std::complex<float>> taps(chan.size());
    std::vector<float> wgt(chan.size()), llr(n_sym * n * bps);
    std::vector<std::uint8_t> hard(bits.size());
    CHECK(orc_zf_taps_f32(bps, reinterpret_cast<const float*>(chan.data()), chan.size(), 40.0f, reinterpret_cast<float*>(taps.data()), wgt.data()) == 0);
    CHECK(orc_demod_c32(n, cp, bps, n_sym, per, reinterpret_cast<const float*>(air.data()), reinterpret_cast<const float*>(taps.data()), wgt.data(), 0.0f, nullptr,
                        llr.data(), hard.data()) == 0);

    auto eq = ofdm::zf_equaliser<Mod>(chan, 40.0f, per);
    CHECK(eq.has_value());
    if (!eq) return;
    CHECK(eq->taps == taps && eq->weights == wgt && eq->sym_per_set == per);
    auto d = ofdm::demod_batch<Mod>(air, n, cp, *eq);
    CHECK(d.has_value());
    if (!d) { std::printf("demod_batch: %s\n", d.error().c_str()); return; }
    CHECK(d->hard == hard);
    CHECK(d->llr == llr);
    std::size_t errors = 0;
    for (std::size_t i = 0; i < bits.size(); ++i) errors += std::size_t(__builtin_popcount(unsigned(bits[i] ^ d->hard[i])));
    CHECK(errors * 50 < bits.size() * 8);                                     // zero forcing at high SNR: nearly every bit comes back
    // taps alone (scalar weight) and weights alone
    ofdm::equaliser only_taps{eq->taps, {}, per}, only_w{{}, eq->weights, per};
    CHECK(orc_demod_c32(n, cp, bps, n_sym, per, reinterpret_cast<const float*>(air.data()), reinterpret_cast<const float*>(taps.data()), nullptr, 1.75f, nullptr,
                        llr.data(), hard.data()) == 0);
    auto dt = ofdm::demod_batch<Mod>(air, n, cp, only_taps, 1.75f);
    CHECK(dt.has_value() && dt->llr == llr && dt->hard == hard);
    CHECK(orc_demod_c32(n, cp, bps, n_sym, per, reinterpret_cast<const float*>(air.data()), nullptr, wgt.data(), 0.0f, nullptr, llr.data(), hard.data()) == 0);
    auto dw = ofdm::demod_batch<Mod>(air, n, cp, only_w);
    CHECK(dw.has_value() && dw->llr == llr && dw->hard == hard);
    // too few sets for the batch, ragged sets: error values, nothing read out of bounds
    ofdm::equaliser few{std::vector<std::complex<float>>(taps.begin(), taps.begin() + n), std::vector<float>(wgt.begin(), wgt.begin() + n), per};
    if (n_sets > 1) { auto bad = ofdm::demod_batch<Mod>(air, n, cp, few); CHECK(!bad.has_value() && bad.error().find("n_sets") != std::string::npos); }
    ofdm::equaliser ragged{std::vector<std::complex<float>>(n + 1), {}, per};
    CHECK(!ofdm::demod_batch<Mod>(air, n, cp, ragged).has_value());
    // an equaliser without arrays is the flat-channel overload
    auto flat_a = ofdm::demod_batch<Mod>(air, n, cp, ofdm::equaliser{}, 0.5f);
    auto flat_b = ofdm::demod_batch<Mod>(air, n, cp, 0.5f);
    CHECK(flat_a.has_value() && flat_b.has_value() && flat_a->llr == flat_b->llr && flat_a->hard == flat_b->hard);
}

static void streaming_pipe()         // sdrlib/pipe.hpp against demod_batch on the same samples
{
    constexpr uint32_t N = 64, CP = 16;
    std::vector<uint8_t> payload(N * 4 * 10 / 8);                  // ten OFDM symbols of 16-QAM
    for (size_t i = 0; i < payload.size(); ++i) payload[i] = uint8_t(i * 29 + 5);
    auto freq = modulation::to_constl<modulation::e16QAM, float>(payload);
    std::vector<std::complex<float>> air;
    CHECK(ofdm::tx_batch(freq, N, CP, air).has_value());
    auto whole = ofdm::demod_batch<modulation::e16QAM>(air, N, CP, 1.5f);
    CHECK(whole.has_value());

    sdrlib::pipe_options o;
    o.n_fft = N; o.cp = CP; o.bits_per_symbol = 4; o.chunk_symbols = 4; o.depth = 2;
    o.llr_to_host = true; o.hard_to_host = true; o.llr_weight = 1.5f;
    auto made = sdrlib::pipe::create(o);
    CHECK(made.has_value());
    if (!made) return;
    sdrlib::pipe p = std::move(*made);
    std::vector<uint8_t> hard;
    std::vector<float> llr;
    auto drain = [&] {
        auto c = p.collect();
        CHECK(c.has_value());
        if (!c) return;
        hard.insert(hard.end(), c->hard.begin(), c->hard.end());
        llr.insert(llr.end(), c->llr.begin(), c->llr.end());
        CHECK(p.release().has_value());
    };
    const size_t per = N + CP;
    for (size_t first = 0; first < 10; first += 4) {               // 4 + 4 + 2 symbols through 2 slots
        const size_t n = std::min<size_t>(4, 10 - first);
        if (p.in_flight() == o.depth) drain();
        auto buf = p.acquire();
        CHECK(buf.has_value() && buf->size() == 4 * per);
        std::copy(air.begin() + first * per, air.begin() + (first + n) * per, buf->begin());
        CHECK(p.submit(n).has_value());
    }
    while (p.in_flight()) drain();
    CHECK(hard == payload && hard == whole->hard);
    CHECK(llr == whole->llr);
    CHECK(!p.collect().has_value());                               // protocol errors are values
    auto bad = sdrlib::pipe::create(sdrlib::pipe_options{.n_fft = 48});
    CHECK(!bad.has_value() && bad.error() == "The sequence size must be of powers of 2");
}

int main(int argc, char** argv)
{
    if (argc > 1 && std::strcmp(argv[1], "--expect-no-device") == 0) {
        std::vector<std::complex<float>> seq{0, 1, 2, 3};
        auto r = fft::fft2(seq.begin(), seq.end());
        CHECK(!r.has_value());
        if (!r) std::printf("error surfaced: %s\n", r.error().c_str());
        std::vector<std::complex<double>> bad{0, 1, 2};
        auto e = fft::fft2(bad.begin(), bad.end());
        CHECK(!e.has_value() && e.error() == "The sequence size must be of powers of 2");
        bool threw = false;
        try { modulation::to_constl<modulation::e16QAM>({1, 2}); } catch (const std::runtime_error&) { threw = true; }
        CHECK(threw);
        std::printf(g_failed ? "FAILED %d checks\n" : "ok (no device)\n", g_failed);
        return g_failed ? 1 : 0;
    }
    fft_roundtrip_double();
    fft_roundtrip_float();
    fft_non_pow2_fails();
    ofdm_roundtrip();
    qam16_hello();
    app_tick();
    extras();
    equalised_demod<modulation::e64QAM>(1024, 256, 30, 14, 0.0005f);          // the north_star shape: tuned tap-holding kernels
    equalised_demod<modulation::e16QAM>(64, 16, 9, 4, 0.002f);                // several symbols per wave
    equalised_demod<modulation::e256QAM>(2048, 512, 5, 2, 0.0002f);
    equalised_demod<modulation::eQPSK>(256, 33, 7, 3, 0.01f);                 // odd prefix: the generic engines
    streaming_pipe();
    std::printf(g_failed ? "FAILED %d checks\n" : "ok: all drop-in checks passed\n", g_failed);
    return g_failed ? 1 : 0;
}
