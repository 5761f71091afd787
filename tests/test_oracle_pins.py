"""Pins the CPU oracle (oracle/sdr_oracle.cpp) against everything the reference holds for the
path: its own gtest cases restated (lib/tests/fft_test.cpp, lib/tests/ofdm_test.cpp), the
reference output recorded at survey time, numpy.fft, and the committed golden
regression vectors.  CPU only."""
import os

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_vectors.npz")


# ---- reference gtest cases, restated ------------------------------------------------------
def test_fft_roundtrip_double(oracle):
    # FFTTest.TransformsSequenceForthAndBackCorrectly / FftIfftForDouble (fft_test.cpp:10-38)
    ref = np.arange(8, dtype=np.complex128)
    rc, f = oracle.fft2(ref)
    assert rc == 0
    rc, back = oracle.fft2(f, inverse=True)
    assert rc == 0
    assert np.all(np.abs(back - ref) < 1e-9)


def test_fft_roundtrip_float(oracle):
    # FFTTest.FftIfftForFloat (fft_test.cpp:40-65): abs <= 1e-5 or rel <= 1e-6 per component
    ref = np.arange(8, dtype=np.complex64)
    _, f = oracle.fft2(ref)
    _, back = oracle.fft2(f, inverse=True)

    def close(x, y):
        d = np.abs(x - y)
        m = np.maximum(np.abs(x), np.abs(y))
        return np.all((d <= 1e-5) | (d <= 1e-6 * m))

    assert close(back.real, ref.real) and close(back.imag, ref.imag)


def test_fft_non_pow2_fails(oracle):
    # FFTTest.SizeNonMultipleOf2Fails (fft_test.cpp:67-72) + the message (fft.hpp:92)
    rc, _ = oracle.fft2(np.array([0, 1, 2], dtype=np.complex128))
    assert rc != 0
    assert oracle.lib().orc_pow2_error().decode() == "The sequence size must be of powers of 2"


def test_fft_sizes_0_and_1_pass(oracle):
    # fft.hpp:91: size 0 and 1 are accepted and untouched
    rc, out = oracle.fft2(np.array([3 + 4j], dtype=np.complex64))
    assert rc == 0 and out[0] == 3 + 4j
    rc, out = oracle.fft2(np.zeros(0, dtype=np.complex64))
    assert rc == 0 and out.size == 0


def test_ofdm_tx_rx_roundtrip(oracle):
    # OFDMTest.TransformsSequenceForthAndBackCorrectly (ofdm_test.cpp:11-26): N = 8, cp = 8
    ref = np.array([0, 1 + 1j, 2 + 2j, -1 - 1j, -2 - 2j, -3 - 3j, 1 - 1j, -1 + 1j], dtype=np.complex128)
    rc, buf = oracle.ofdm_tx(ref, 8, 8)
    assert rc == 0 and buf.shape == (1, 16)
    assert np.array_equal(buf[0, :8], buf[0, 8:])          # cp == N: the prefix is the whole symbol
    rc, res = oracle.ofdm_rx(buf, 8, 8)
    assert rc == 0
    assert np.all(np.abs(res[0] - ref) < 1e-9)


def test_qam16_hello_roundtrip(oracle):
    # OFDMTest.Transforms16QAMForthAndBackCorrectly (ofdm_test.cpp:28-36), T = double default
    data = np.frombuffer(b"Hello", dtype=np.uint8)
    sym = oracle.to_constl(4, data, dtype=np.complex128)
    assert sym.size == 10
    assert oracle.from_constl(4, sym).tobytes() == b"Hello"
    symf = oracle.to_constl(4, data, dtype=np.complex64)
    assert oracle.from_constl(4, symf).tobytes() == b"Hello"


def test_qam16_table_matches_reference_listing(oracle):
    # modulation.hpp:29-47, index -> point
    expected = [-3 - 3j, -3 - 1j, -3 + 3j, -3 + 1j, -1 - 3j, -1 - 1j, -1 + 3j, -1 + 1j,
                3 - 3j, 3 - 1j, 3 + 3j, 3 + 1j, 1 - 3j, 1 - 1j, 1 + 3j, 1 + 1j]
    nrm, inrm, tab = oracle.constellation(4)
    assert list(tab) == expected
    assert nrm == np.float32(1.0) / np.sqrt(np.float32(10.0)) and inrm == np.sqrt(np.float32(10.0))
    # 'H' = 0x48 -> symbols table[4]*norm, table[8]*norm (high nibble first, modulation.hpp:87-91)
    s = oracle.to_constl(4, np.array([0x48], dtype=np.uint8))
    assert s[0] == np.complex64(complex(-1 * nrm, -3 * nrm)) and s[1] == np.complex64(complex(3 * nrm, -3 * nrm))


def test_from_constl_drops_unpaired_symbol(oracle):
    # modulation.hpp:103
    sym = oracle.to_constl(4, np.array([0xAB, 0xCD], dtype=np.uint8))
    assert oracle.from_constl(4, sym[:3]).tobytes() == b"\xab"


def test_nearest_nan_and_ties(oracle):
    # NaN never compares '<' -> index 0 (modulation.hpp:60); exact ties -> lowest index
    nrm, _, tab = oracle.constellation(4)
    pts = np.array([complex(np.nan, 0), complex(0, np.nan), 0j, complex(np.inf, 0)], dtype=np.complex64)
    idx = oracle.nearest(4, pts)
    assert idx[0] == 0 and idx[1] == 0 and idx[3] == 0
    assert idx[2] == 5          # (0,0): -1-1j (5), -1+1j (7), 1-1j (13), 1+1j (15) tie; lowest wins


# ---- survey-time reference output and numpy cross-checks -----------------------------------
def test_fft_0_to_7_float_matches_recorded_reference_output(oracle):
    # reference fft2({0..7}) in float = 28, -4+9.657i, -4+4i, -4+1.657i, -4, ...
    # with the recurrence error visible in bin 1: -3.99999952
    _, f = oracle.fft2(np.arange(8, dtype=np.complex64))
    assert f[0] == 28
    assert f.real[1] == np.float32(-3.99999952)
    assert np.allclose(f, [28, -4 + 9.65685425j, -4 + 4j, -4 + 1.65685425j, -4, -4 - 1.65685425j, -4 - 4j, -4 - 9.65685425j], atol=2e-6)


@pytest.mark.parametrize("n", [2, 4, 8, 64, 256, 1024, 2048, 4096])
def test_fft_against_numpy(oracle, n):
    rng = np.random.default_rng(n)
    x = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    _, f64 = oracle.fft2(x.astype(np.complex128))
    ref = np.fft.fft(x)
    assert np.abs(f64 - ref).max() <= 1e-11 * np.abs(ref).max()
    _, f32 = oracle.fft2(x.astype(np.complex64))
    assert np.abs(f32 - ref).max() <= 5e-6 * np.abs(ref).max()      # 2.2e-6 at 4096
    _, i64 = oracle.fft2(x.astype(np.complex128), inverse=True)
    assert np.abs(i64 - np.fft.ifft(x)).max() <= 1e-11 * np.abs(x).max()


def test_twiddle_table_reproduces_fft(oracle):
    # the per-stage table is exactly the sequence the in-loop recurrence uses
    n = 256
    tw = oracle.twiddles(n)
    rng = np.random.default_rng(5)
    x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
    rev = np.array([int(format(i, "08b")[::-1], 2) for i in range(n)])
    a = x[rev].copy()
    half = 1
    f32 = np.float32
    while half < n:
        for blk in range(0, n, 2 * half):
            for j in range(half):
                w = tw[half - 1 + j]
                e, o_ = a[blk + j], a[blk + j + half]
                tr = f32(f32(o_.real * w.real) - f32(o_.imag * w.imag))
                ti = f32(f32(o_.real * w.imag) + f32(o_.imag * w.real))
                a[blk + j] = complex(f32(e.real + tr), f32(e.imag + ti))
                a[blk + j + half] = complex(f32(e.real - tr), f32(e.imag - ti))
        half *= 2
    _, ref = oracle.fft2(x)
    assert np.array_equal(a, ref)


def test_philox_known_answers(oracle):
    # Random123 kat_vectors, philox4x32-10
    assert list(oracle.philox([0, 0, 0, 0], [0, 0])) == [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]
    assert list(oracle.philox([0xFFFFFFFF] * 4, [0xFFFFFFFF] * 2)) == [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]
    assert list(oracle.philox([0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344], [0xA4093822, 0x299F31D0])) == \
        [0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1]


def test_gaussian_generator_functions(oracle):
    """X4 (no reference counterpart): the generator's own ln / sincos / Box-Muller against numpy, its tail bound
    (32-bit uniform: |z| <= sqrt(64 ln 2) sigma = 6.66 sigma) and its edge words."""
    us = np.concatenate([2.0 ** -np.arange(0, 33), np.random.default_rng(1).uniform(2.0 ** -32, 1.0, 4000), 1.0 - 2.0 ** -np.arange(1, 25)]).astype(np.float32)
    got = np.array([oracle.ln_unit(u) for u in us])
    assert np.abs(got - np.log(us.astype(np.float64))).max() < 4e-6
    ks = np.concatenate([np.arange(0, 1 << 24, 65536 // 4 + 1), [0, 1, 0x7FFF, 0x8000, 0x8001, 0xFFFFFF, 0xFF8000, 0xFF7FFF]])
    sc = np.array([oracle.sincos_turn(int(k)) for k in ks])
    ang = 2 * np.pi * ks / float(1 << 24)
    assert np.abs(sc[:, 0] - np.sin(ang)).max() < 3e-7 and np.abs(sc[:, 1] - np.cos(ang)).max() < 3e-7
    # edge words: the largest radius, the zero radius (never NaN)
    re, im = oracle.gaussian_pair(0, 0)
    assert abs(np.hypot(re, im) - np.sqrt(64 * np.log(2.0))) < 1e-5 and im == 0.0
    for x2 in (0, 0x40000000, 0xFFFFFFFF, 0x12345678):
        re, im = oracle.gaussian_pair(0xFFFFFFFF, x2)
        assert re == 0.0 and im == 0.0
        re, im = oracle.gaussian_pair(0xFFFFFF00, x2)
        assert np.isfinite(re) and np.isfinite(im) and np.hypot(re, im) < 1e-3
    # moments of 2 x 200000 samples from Philox words
    rng = np.random.default_rng(2)
    w = rng.integers(0, 1 << 32, (200000, 2), dtype=np.uint64)
    z = np.array([oracle.gaussian_pair(int(a), int(b)) for a, b in w]).ravel()
    assert np.isfinite(z).all() and np.abs(z).max() <= 6.67
    assert abs(z.mean()) < 0.01 and abs(z.var() - 1.0) < 0.01 and abs((z ** 4).mean() - 3.0) < 0.06
    t = oracle.rng_tables()
    assert t.size == 576 and t[64] == 0.0 and t[65] == 1.0
    assert np.abs(t[1:64:2] + np.log(t[0:64:2].astype(np.float64))).max() < 4e-8      # L_i = fl(-ln r_i): half an ulp of 0.69


# ---- extensions: internal consistency -------------------------------------------------------
@pytest.mark.parametrize("bps", [1, 2, 4, 6, 8, 10])
def test_constellation_unit_energy_and_roundtrip(oracle, bps):
    nrm, inrm, tab = oracle.constellation(bps)
    assert abs(np.mean(np.abs(tab * nrm) ** 2) - 1.0) < 1e-6
    rng = np.random.default_rng(bps)
    data = rng.integers(0, 256, 30 * bps, dtype=np.uint8)
    sym = oracle.to_constl(bps, data)
    assert np.array_equal(oracle.from_constl(bps, sym), data)
    # Gray: horizontally / vertically adjacent points differ in exactly one bit
    idx = np.arange(1 << bps)
    for a in idx:
        for b in idx:
            if a < b and abs(tab[a] - tab[b]) == 2:
                assert bin(a ^ b).count("1") == 1


@pytest.mark.parametrize("bps", [2, 4, 6, 8])
def test_llr_sign_agrees_with_hard_decision_off_boundaries(oracle, bps):
    rng = np.random.default_rng(100 + bps)
    pts = ((rng.standard_normal(4000) + 1j * rng.standard_normal(4000)) * 0.8).astype(np.complex64)
    idx = oracle.nearest(bps, pts)
    llr = oracle.soft_demap(bps, pts, 2.0)
    bits = (idx[:, None] >> (bps - 1 - np.arange(bps))) & 1
    ok = (llr < 0) == (bits == 1)
    assert ok[np.abs(llr) > 1e-4].all()


def test_llr_matches_float64_formula(oracle):
    rng = np.random.default_rng(7)
    for bps in (2, 4, 6, 8):
        nrm, inrm, tab = oracle.constellation(bps)
        pts = ((rng.standard_normal(500) + 1j * rng.standard_normal(500)) * 0.7).astype(np.complex64)
        llr = oracle.soft_demap(bps, pts, 1.5)
        u = pts.astype(np.complex128) * float(inrm)
        d2 = np.abs(u[:, None] - tab[None, :].astype(np.complex128)) ** 2
        for b in range(bps):
            mask = ((np.arange(1 << bps) >> (bps - 1 - b)) & 1).astype(bool)
            ref = (d2[:, mask].min(1) - d2[:, ~mask].min(1)) * 1.5
            assert np.abs(llr[:, b] - ref).max() <= 1e-4 * np.abs(ref).max()


def test_link_ber_close_to_theory(oracle):
    from sdr_amd.link import noise_sigma, llr_weight, theory_ber
    for bps, ebn0 in ((2, 4.0), (4, 8.0)):
        c = oracle.link_ber(64, 16, bps, 11, 0, 1500, noise_sigma(64, bps, ebn0), llr_weight(bps, ebn0))
        ber = c[0] / c[1]
        th = theory_ber(bps, ebn0)
        assert 0.8 * th < ber < 1.25 * th, (bps, ber, th)
        assert c[1] == 1500 * 64 * bps and c[4] == 1500


def test_generator_is_split_invariant(oracle):
    a, ba = oracle.link_generate(64, 16, 4, 3, 0, 6, 0.02)
    b1, bb1 = oracle.link_generate(64, 16, 4, 3, 0, 2, 0.02)
    b2, bb2 = oracle.link_generate(64, 16, 4, 3, 2, 4, 0.02)
    assert np.array_equal(a, np.concatenate([b1, b2])) and np.array_equal(ba, np.concatenate([bb1, bb2]))


# ---- committed golden regression vectors ----------------------------------------------------
def test_golden_vectors(oracle):
    g = np.load(GOLDEN)
    for n in (8, 64, 1024):
        assert np.array_equal(oracle.fft2(g[f"fft_in_c64_{n}"])[1], g[f"fft_out_c64_{n}"])
        assert np.array_equal(oracle.fft2(g[f"fft_in_c64_{n}"], inverse=True)[1], g[f"ifft_out_c64_{n}"])
        assert np.array_equal(oracle.fft2(g[f"fft_in_c128_{n}"])[1], g[f"fft_out_c128_{n}"])
    assert np.array_equal(oracle.twiddles(1024), g["twiddles_c64_1024"])
    assert np.array_equal(oracle.twiddles(64, inverse=True), g["twiddles_inv_c64_64"])
    for bps in (2, 4, 6, 8):
        pts = g[f"demap_pts_{bps}"]
        assert np.array_equal(oracle.nearest(bps, pts), g[f"demap_idx_{bps}"])
        assert np.array_equal(oracle.soft_demap(bps, pts, 1.75), g[f"demap_llr_{bps}"], equal_nan=True)
    rx, bits = oracle.link_generate(64, 16, 2, 1, 0, 3, 0.05)
    assert np.array_equal(rx, g["link_rx_64_qpsk"]) and np.array_equal(bits, g["link_bits_64_qpsk"])
    assert np.array_equal(oracle.link_ber(1024, 256, 4, 2, 28, 2, 0.012, 9.0), g["link_ber_1024_16qam"])


def test_threaded_demod_equals_single_thread(oracle):
    """bench.py's all-threads CPU leg: symbols split over host threads, same outputs."""
    rng = np.random.default_rng(3)
    n, cp, bps, n_sym, per = 64, 16, 6, 45, 7
    n_sets = (n_sym + per - 1) // per
    rx, _ = oracle.link_generate(n, cp, bps, 9, 0, n_sym, np.float32(0.02))
    eq = (rng.standard_normal((n_sets, n)) + 1j * rng.standard_normal((n_sets, n))).astype(np.complex64)
    wgt = rng.random((n_sets, n), dtype=np.float32)
    one = oracle.demod(rx, n, cp, bps, eq=eq, weight=wgt, sym_per_set=per)
    for threads in (2, 3, 8):
        many = oracle.demod(rx, n, cp, bps, eq=eq, weight=wgt, sym_per_set=per, threads=threads)
        for key in ("fft", "llr", "hard"):
            assert np.array_equal(one[key], many[key]), (threads, key)
