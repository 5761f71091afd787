"""GPU parity tests: the HIP path (through the C ABI of libsdr_amd.so) against the CPU oracle on
identical seeded inputs.  Bars: bit-exact for hard decisions, packed bits, indices and counters;
FFT outputs and LLRs within 1e-4 of the per-symbol maximum magnitude (north_star), and -- since
the kernels evaluate the reference's own operation DAG -- exact equality is additionally
required wherever the test says so."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_vectors.npz")
TOL = 1e-4


def dev(a):
    return torch.from_numpy(np.array(a, copy=True, order="C")).cuda()


def rel_err(a, ref):
    a = np.asarray(a).reshape(ref.shape[0], -1)
    r = np.asarray(ref).reshape(ref.shape[0], -1)
    scale = np.abs(r).max(axis=1, keepdims=True)
    scale[scale == 0] = 1.0
    return float((np.abs(a - r) / scale).max())


def same(a, b, **kw):
    return bool(np.array_equal(np.asarray(a), np.asarray(b), **kw))


def rand_c(rng, shape, dtype=np.complex64):
    return (rng.standard_normal(shape) + 1j * rng.standard_normal(shape)).astype(dtype)


# ---- reference gtest cases through the device path ------------------------------------------
def test_reference_fft_roundtrip_cases(ctx):
    for dt, tol in ((torch.complex128, 1e-9), (torch.complex64, 1e-5)):
        ref = torch.arange(8, device="cuda").to(dt)
        x = ref.clone()
        ctx.fft2_(x)
        ctx.ifft2_(x)
        assert (x - ref).abs().max().item() < tol           # fft_test.cpp:10-65


def test_reference_non_pow2_fails(ctx):
    from sdr_amd.capi import SdrError, SDR_ERR_NOT_POW2
    x = torch.zeros(3, dtype=torch.complex128, device="cuda")
    with pytest.raises(SdrError) as e:
        ctx.fft2_(x)                                          # fft_test.cpp:67-72
    assert e.value.code == SDR_ERR_NOT_POW2 and e.value.message == "The sequence size must be of powers of 2"


def test_reference_ofdm_roundtrip_case(ctx):
    ref = np.array([0, 1 + 1j, 2 + 2j, -1 - 1j, -2 - 2j, -3 - 3j, 1 - 1j, -1 + 1j], dtype=np.complex128)
    buf = ctx.ofdm_tx(dev(ref), 8, 8)                         # ofdm_test.cpp:11-26
    res = ctx.ofdm_rx(buf, 8, 8).cpu().numpy()
    assert np.all(np.abs(res[0] - ref) < 1e-9)


def test_reference_hello_case(ctx):
    data = np.frombuffer(b"Hello", dtype=np.uint8)
    for dt in (torch.complex128, torch.complex64):
        sym = ctx.to_constl(dev(data), "16qam", dtype=dt)
        assert sym.numel() == 10
        assert ctx.from_constl(sym, "16qam").cpu().numpy().tobytes() == b"Hello"      # ofdm_test.cpp:28-36


# ---- fft2 / ifft2: exact DAG parity ------------------------------------------------------
@pytest.mark.parametrize("n", [1, 2, 4, 8, 16, 32, 64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384])
def test_fft2_c32_bit_exact(ctx, oracle, n):
    rng = np.random.default_rng(n)
    batch = 5 if n <= 4096 else 2
    x = rand_c(rng, (batch, n))
    for inverse in (False, True):
        rc, ref = oracle.fft2(x, inverse)
        assert rc == 0
        got = ctx.fft2_(dev(x), inverse).cpu().numpy()
        assert same(got, ref), (n, inverse, rel_err(got, ref))


@pytest.mark.parametrize("n", [2, 8, 64, 1024, 4096, 8192])
def test_fft2_c64_bit_exact(ctx, oracle, n):
    rng = np.random.default_rng(1000 + n)
    x = rand_c(rng, (3, n), np.complex128)
    for inverse in (False, True):
        rc, ref = oracle.fft2(x, inverse)
        got = ctx.fft2_(dev(x), inverse).cpu().numpy()
        assert same(got, ref), (n, inverse, rel_err(got, ref))


def test_fft2_long_sequences_use_global_path(ctx, oracle):
    rng = np.random.default_rng(77)
    for n, dt in ((32768, np.complex64), (16384, np.complex128)):
        x = rand_c(rng, (2, n), dt)
        rc, ref = oracle.fft2(x)
        got = ctx.fft2_(dev(x)).cpu().numpy()
        assert same(got, ref)
        rc, ref = oracle.fft2(x, True)
        got = ctx.fft2_(dev(x), True).cpu().numpy()
        assert same(got, ref)


def test_fft2_empty_and_unit(ctx):
    x = torch.tensor([3 + 4j], dtype=torch.complex64, device="cuda")
    assert ctx.fft2_(x).cpu()[0] == 3 + 4j                   # fft.hpp:91
    ctx.fft2_(torch.zeros((0,), dtype=torch.complex64, device="cuda"))


def test_fft2_golden(ctx):
    g = np.load(GOLDEN)
    for n in (8, 64, 1024):
        assert same(ctx.fft2_(dev(g[f"fft_in_c64_{n}"])).cpu().numpy(), g[f"fft_out_c64_{n}"])
        assert same(ctx.fft2_(dev(g[f"fft_in_c64_{n}"]), True).cpu().numpy(), g[f"ifft_out_c64_{n}"])
        assert same(ctx.fft2_(dev(g[f"fft_in_c128_{n}"])).cpu().numpy(), g[f"fft_out_c128_{n}"])


# ---- ofdm::rx / tx batched -------------------------------------------------------------------
@pytest.mark.parametrize("n,cp", [(8, 8), (64, 16), (256, 64), (1024, 256), (2048, 512), (4096, 1024), (128, 0), (512, 37), (128, 32), (512, 128),
                                  (16, 4), (8192, 2048), (1024, 1024), (1024, 37), (2048, 1), (8192, 3), (4096, 5)])
@pytest.mark.parametrize("dt", [np.complex64, np.complex128])
def test_ofdm_rx_tx_bit_exact(ctx, oracle, n, cp, dt):
    """fp32 and fp64 (fft_test.cpp:12-22 and ofdm_test.cpp:13-25 run the reference in double)."""
    rng = np.random.default_rng(n + cp)
    s = 7
    rx = rand_c(rng, (s, n + cp), dt)
    rc, ref = oracle.ofdm_rx(rx, n, cp)
    got = ctx.ofdm_rx(dev(rx), n, cp).cpu().numpy()
    assert same(got, ref), rel_err(got, ref)
    fr = rand_c(rng, (s, n), dt)
    rc, ref = oracle.ofdm_tx(fr, n, cp)
    got = ctx.ofdm_tx(dev(fr), n, cp).cpu().numpy()
    assert got.dtype == dt and same(got, ref), rel_err(got, ref)
    # round trip property at any size
    back = ctx.ofdm_rx(ctx.ofdm_tx(dev(fr), n, cp), n, cp).cpu().numpy()
    assert rel_err(back, fr) < (1e-5 if dt == np.complex64 else 1e-13)


def _random_tx_cases(count=32, seed=4242):
    rng = np.random.default_rng(seed)
    return [(i, int(rng.choice([16, 64, 128, 256, 512, 1024, 2048, 4096, 8192])), int(rng.integers(1, 20)), int(rng.integers(0, 3)))
            for i in range(count)]


@pytest.mark.parametrize("dt", [np.complex64, np.complex128])
@pytest.mark.parametrize("i,n,n_sym,shift", _random_tx_cases())
def test_ofdm_tx_rx_random_layouts(ctx, oracle, i, n, n_sym, shift, dt):
    """Batched ofdm::tx / ofdm::rx with random prefix lengths (odd ones included), batch sizes and
    buffers that start 0, 1 or 2 elements into an allocation: every dispatch path equals the oracle,
    in fp32 and in fp64."""
    rng = np.random.default_rng(500 + i)
    cp = int(rng.integers(0, n + 1))
    fr = rand_c(rng, (n_sym, n), dt)
    tdt = torch.complex64 if dt == np.complex64 else torch.complex128
    pool = torch.empty(n_sym * n + 4, dtype=tdt, device="cuda")
    src = pool[shift: shift + n_sym * n].view(n_sym, n)
    src.copy_(dev(fr))
    rc, ref = oracle.ofdm_tx(fr, n, cp)
    tx = ctx.ofdm_tx(src, n, cp)
    assert same(tx.cpu().numpy(), ref)
    rpool = torch.empty(n_sym * (n + cp) + 4, dtype=tdt, device="cuda")
    rsrc = rpool[shift: shift + n_sym * (n + cp)].view(n_sym, n + cp)
    rsrc.copy_(tx)
    rc, rref = oracle.ofdm_rx(ref, n, cp)
    assert same(ctx.ofdm_rx(rsrc, n, cp).cpu().numpy(), rref)


# ---- modulation -------------------------------------------------------------------------------
@pytest.mark.parametrize("bps", [1, 2, 4, 6, 8, 10])
def test_to_from_constl_bit_exact(ctx, oracle, bps):
    rng = np.random.default_rng(bps)
    data = rng.integers(0, 256, 15 * bps, dtype=np.uint8)
    for dt, npdt in ((torch.complex64, np.complex64), (torch.complex128, np.complex128)):
        sym = ctx.to_constl(dev(data), bps, dtype=dt)
        ref = oracle.to_constl(bps, data, dtype=npdt)
        assert same(sym.cpu().numpy(), ref)
        assert same(ctx.from_constl(sym, bps).cpu().numpy(), data)
    # ragged: symbol counts that do not fill a byte (modulation.hpp:103 drops the tail)
    sym = oracle.to_constl(bps, data)
    for n_sym in (1, 3, 5, 7):
        got = ctx.from_constl(dev(sym[:n_sym]), bps).cpu().numpy()
        assert same(got, oracle.from_constl(bps, sym[:n_sym]))


@pytest.mark.parametrize("bps", [2, 4, 6, 8])
def test_nearest_and_llr_on_boundaries_and_outliers(ctx, oracle, bps):
    g = np.load(GOLDEN)
    pts = g[f"demap_pts_{bps}"]
    idx = ctx.nearest(dev(pts), bps).cpu().numpy().astype(np.uint16)
    assert same(idx, g[f"demap_idx_{bps}"])
    llr = ctx.soft_demap(dev(pts), bps, 1.75).cpu().numpy()
    assert same(llr, g[f"demap_llr_{bps}"], equal_nan=True)


@pytest.mark.parametrize("bps", [1, 2, 4, 6, 8, 10])
@pytest.mark.parametrize("dt", [np.float32, np.float64])
def test_nearest_adversarial_points_bit_exact(ctx, oracle, bps, dt):
    """Everything that can separate a slicer from the reference's scan (modulation.hpp:49-67): one
    coordinate within a few ulps of every decision boundary and of every level while the other is
    tiny, ordinary, huge (its square absorbs the difference), on a level, infinite or NaN; both
    orders; float and double.  The device's scan-free path must fall back exactly when it has to."""
    mi = (bps + 1) // 2 if bps > 1 else 1
    levels = 1 << mi
    inorm = np.sqrt(dt(2.0 * (levels * levels - 1) / 3.0)) if bps > 1 else dt(1.0)
    marks = np.arange(-(levels - 1) - 6, levels + 6, dtype=np.float64)             # levels, boundaries, beyond the edge
    # ... and the margins around the boundaries inside which the threshold decision defers to the scan
    # (round 3: the slicer's reach test |re| + |im| <= (L - 1) + 62 grid units replaced the per-axis one at 28 beyond the edge)
    marks = np.concatenate([marks] + [marks + o for o in (2.0 ** -12, -2.0 ** -12, 2.0 ** -10, -2.0 ** -10, 2.0 ** -13, -2.0 ** -13, 28.0, -28.0, 28.5, -28.5,
                                                             56.0, -56.0, 62.0, -62.0, 62.5, -62.5)])
    near = []
    for v in marks:
        x = dt(v) / inorm
        for _ in range(4):
            near.append(x)
            x = np.nextafter(x, dt(np.inf))
        x = dt(v) / inorm
        for _ in range(4):
            x = np.nextafter(x, dt(-np.inf))
            near.append(x)
    near = np.array(near, dtype=dt)
    other = np.array([0.0, 1e-30, -1e-30, 0.37, -0.91, 1.0, 3.0, 40.0, -2.5e3, 1e6, -1e9, 3e18, np.inf, -np.inf, np.nan], dtype=dt)
    other = np.concatenate([other, (np.arange(-(levels - 1), levels, 2, dtype=np.float64) / float(inorm)).astype(dt)])
    # partners far out but still inside (or just outside) the reach of the slicer: their squares are where one ulp of the sum is largest
    far = np.array([levels - 1 + o for o in (20.0, 45.0, 55.0, 60.0, 61.0, 61.9, 62.0, 62.1, 64.0)])
    other = np.concatenate([other, (far / float(inorm)).astype(dt), (-far / float(inorm)).astype(dt)])
    a, b = np.meshgrid(near, other, indexing="ij")
    cdt = np.complex64 if dt is np.float32 else np.complex128
    pts = np.empty(2 * a.size, dtype=cdt)                       # assembled by parts: 1j * inf would make NaNs
    pts.real = np.concatenate([a.ravel(), b.ravel()])
    pts.imag = np.concatenate([b.ravel(), a.ravel()])
    with np.errstate(all="ignore"):
        want = oracle.nearest(bps, pts)
    got = ctx.nearest(dev(pts), bps).cpu().numpy().astype(np.uint16)
    assert same(got, want.astype(np.uint16))
    # and through from_constl (whole bytes only)
    k = (pts.size // 8) * 8
    with np.errstate(all="ignore"):
        want_bytes = oracle.from_constl(bps, pts[:k])
    assert same(ctx.from_constl(dev(pts[:k]), bps).cpu().numpy(), want_bytes)


@pytest.mark.parametrize("bps", [2, 6, 8, 10])
@pytest.mark.parametrize("n_sym", [1, 63, 64, 65, 257, 2048 + 5, (1 << 20) + 77])
def test_soft_demap_ragged_sizes(ctx, oracle, bps, n_sym):
    """Soft demapping at sizes that leave the last wave partly empty, in a grid that is / is not a
    multiple of the XCD count, with nothing written behind the last LLR."""
    rng = np.random.default_rng(n_sym + bps)
    pts = rand_c(rng, n_sym) * np.float32(0.8)
    want = oracle.soft_demap(bps, pts, 1.25)
    guard = torch.full((n_sym * bps + 64,), -3.5, dtype=torch.float32, device="cuda")
    got = ctx.soft_demap(dev(pts), bps, 1.25, out=guard[: n_sym * bps].view(n_sym, bps))
    assert same(got.cpu().numpy().reshape(want.shape), want)
    assert bool((guard[n_sym * bps:] == -3.5).all())


@pytest.mark.parametrize("bps", [1, 2, 4, 6, 8, 10])
def test_nearest_random_noise_bit_exact(ctx, oracle, bps):
    rng = np.random.default_rng(40 + bps)
    pts = rand_c(rng, 200000) * np.float32(0.8)
    pts[:4] = [complex(np.nan, 0), complex(0, np.nan), 0j, complex(np.inf, 1)]
    pts[4:2000] *= np.float32(40.0)                            # far outliers: rounding-created ties
    idx = ctx.nearest(dev(pts), bps).cpu().numpy().astype(np.uint16)
    assert same(idx, oracle.nearest(bps, pts))
    if bps in (2, 4):
        p64 = pts.astype(np.complex128)
        got = ctx.from_constl(dev(p64[: 8 * 1000]), bps).cpu().numpy()
        assert same(got, oracle.from_constl(bps, p64[: 8 * 1000]))
    llr = ctx.soft_demap(dev(pts[4:]), bps, 0.37).cpu().numpy()
    assert same(llr, oracle.soft_demap(bps, pts[4:], 0.37))


def test_zf_taps(ctx, oracle):
    rng = np.random.default_rng(8)
    h = rand_c(rng, 4096) * np.float32(0.7)
    taps, w = ctx.zf_taps(dev(h), 6, 17.5)
    rt, rw = oracle.zf_taps(6, h, 17.5)
    assert same(taps.cpu().numpy(), rt) and same(w.cpu().numpy(), rw)


# ---- fused receive path ------------------------------------------------------------------------
CASES = [(64, 16, 2), (64, 16, 4), (128, 32, 6), (256, 64, 8), (512, 128, 4), (1024, 256, 4), (1024, 256, 6),
         (2048, 512, 6), (4096, 1024, 8), (1024, 256, 2), (4096, 1024, 2), (2048, 0, 8), (32, 8, 4), (1024, 100, 1), (256, 64, 10), (1024, 256, 10), (4096, 1024, 10), (64, 16, 10), (8192, 2048, 4), (8192, 2048, 6), (8192, 0, 10)]


@pytest.mark.parametrize("n,cp,bps", CASES)
def test_demod_parity(ctx, oracle, n, cp, bps):
    from sdr_amd.link import noise_sigma, llr_weight
    n_sym = 29 if n <= 1024 else 9
    ebn0 = {1: 4.0, 2: 5.0, 4: 9.0, 6: 14.0, 8: 19.0, 10: 24.0}[bps]
    sigma = np.float32(noise_sigma(n, bps, ebn0))
    w = np.float32(llr_weight(bps, ebn0))
    rx, bits = oracle.link_generate(n, cp, bps, 1234 + n, 5, n_sym, sigma)
    ref = oracle.demod(rx, n, cp, bps, weight=w)
    counters = torch.zeros(8, dtype=torch.int64, device="cuda")
    out = ctx.demod(dev(rx), n, cp, bps, llr_weight=float(w), want_llr=True, want_hard=True, want_fft=True,
                    ref_hard=dev(bits), counters=counters)
    assert same(out.hard.cpu().numpy(), ref["hard"])                 # hard decisions: bit exact
    assert rel_err(out.fft.cpu().numpy(), ref["fft"]) <= TOL                    # FFT: 1e-4 relative
    assert same(out.fft.cpu().numpy(), ref["fft"])                    # ... and in fact identical
    got_llr = out.llr.cpu().numpy()
    assert rel_err(got_llr, ref["llr"]) <= TOL                                  # LLR: 1e-4 relative
    assert same(got_llr, ref["llr"])
    c = counters.cpu().numpy()
    assert c[0] == oracle.count_bit_errors(ref["hard"], bits) and c[0] > 0      # error count: bit exact
    assert c[1] == n_sym * n * bps and c[3] == n_sym * n and c[4] == n_sym
    # the other counters, restated with numpy on the oracle's outputs
    tx = np.unpackbits(bits).reshape(-1, bps)
    rxb = np.unpackbits(ref["hard"]).reshape(-1, bps)
    assert c[2] == int((tx != rxb).any(axis=1).sum())
    l = ref["llr"].reshape(-1, bps)
    q = oracle.llr_abs_units(l)
    assert c[5] == int(q.sum())
    assert c[6] == int(((l < 0).astype(np.uint8) != tx).sum())
    # a second call accumulates
    ctx.demod(dev(rx), n, cp, bps, llr_weight=float(w), want_llr=False, ref_hard=dev(bits), counters=counters)
    assert same(counters.cpu().numpy(), 2 * c)


@pytest.mark.parametrize("n,cp,bps", [(64, 16, 2), (1024, 256, 4), (2048, 512, 6), (4096, 1024, 8), (8192, 2048, 6), (2048, 512, 10), (4096, 1024, 2), (8192, 2048, 1)])
def test_demod_with_equaliser_sets(ctx, oracle, n, cp, bps):
    from sdr_amd.link import noise_sigma
    rng = np.random.default_rng(n)
    n_sym, per = 30, 14
    n_sets = (n_sym + per - 1) // per
    h = (rand_c(rng, (n_sets, n)) * np.float32(np.sqrt(0.5))).astype(np.complex64)
    h[np.abs(h) < 0.1] += np.float32(0.3)
    sigma = np.float32(noise_sigma(n, bps, 30.0))
    rx, bits = oracle.link_generate(n, cp, bps, 99, 0, n_sym, sigma, chan=h, sym_per_set=per)
    taps, wgt = oracle.zf_taps(bps, h, 50.0)
    ref = oracle.demod(rx, n, cp, bps, eq=taps, weight=wgt, sym_per_set=per)
    dtaps, dw = ctx.zf_taps(dev(h.ravel()), bps, 50.0)
    out = ctx.demod(dev(rx), n, cp, bps, eq=dtaps, llr_weight=dw, sym_per_set=per, want_hard=True, want_fft=True)
    assert same(out.hard.cpu().numpy(), ref["hard"])
    assert same(out.fft.cpu().numpy(), ref["fft"])
    assert rel_err(out.llr.cpu().numpy(), ref["llr"]) <= TOL
    assert same(out.llr.cpu().numpy(), ref["llr"])
    # high SNR + zero forcing: almost every bit must come back
    assert oracle.count_bit_errors(ref["hard"], bits) < 0.02 * bits.size * 8


def _random_demod_cases(count=64, seed=20261111):
    # SDR_SOAK_COUNT / SDR_SOAK_SEED: a longer sweep from another seed (tools/soak.sh), same checks
    count = int(os.environ.get("SDR_SOAK_COUNT", count))
    seed = int(os.environ.get("SDR_SOAK_SEED", seed))
    rng = np.random.default_rng(seed)
    cases = []
    for i in range(count):
        n = int(rng.choice([64, 128, 256, 512, 1024, 2048, 4096, 8192]))
        cp = int(rng.choice([0, n // 4, 2 * int(rng.integers(0, n // 2 + 1))]))
        bps = int(rng.choice([1, 2, 4, 6, 8, 10]))
        n_sym = int(rng.integers(1, 36 if n <= 1024 else 12))
        taps = str(rng.choice(["none", "none", "both", "taps", "weights"]))
        outs = int(rng.integers(1, 16))                  # bit 0 llr, 1 hard, 2 bins, 3 counters (needs hard reference)
        cases.append((i, n, cp, bps, n_sym, taps, outs))
    return cases


@pytest.mark.parametrize("i,n,cp,bps,n_sym,taps,outs", _random_demod_cases())
def test_demod_random_configurations(ctx, oracle, i, n, cp, bps, n_sym, taps, outs):
    """Seeded sweep over size x prefix x order x batch x taps/weights x requested outputs: whichever
    instantiation the dispatch picks, every output it produces equals the oracle's."""
    from sdr_amd.link import noise_sigma
    rng = np.random.default_rng(1000 + i)
    per = int(rng.integers(1, 9))
    n_sets = (n_sym + per - 1) // per
    h = (rand_c(rng, (n_sets, n)) * np.float32(np.sqrt(0.5))).astype(np.complex64)
    h[np.abs(h) < 0.1] += np.float32(0.3)
    use_chan = taps in ("both", "taps")
    ebn0 = {1: 6.0, 2: 7.0, 4: 11.0, 6: 16.0, 8: 21.0, 10: 26.0}[bps]
    # first symbol on a set boundary: the generator's channel sets then line up with the demodulator's
    rx, bits = oracle.link_generate(n, cp, bps, 77 + i, i * per, n_sym, np.float32(noise_sigma(n, bps, ebn0)),
                                    chan=(h if use_chan else None), sym_per_set=(per if use_chan else 0))
    t, w = oracle.zf_taps(bps, h, 9.0)
    eq = t if taps in ("both", "taps") else None
    wt = w if taps in ("both", "weights") else np.float32(0.8)
    sps = per if taps != "none" else 0
    want_llr, want_hard, want_fft, want_cnt = bool(outs & 1), bool(outs & 2), bool(outs & 4), bool(outs & 8)
    ref = oracle.demod(rx, n, cp, bps, eq=eq, weight=wt, sym_per_set=sps)
    counters = torch.zeros(8, dtype=torch.int64, device="cuda") if want_cnt else None
    out = ctx.demod(dev(rx), n, cp, bps, eq=(dev(eq.ravel()) if eq is not None else None),
                    llr_weight=(dev(wt.ravel()) if isinstance(wt, np.ndarray) else float(wt)), sym_per_set=sps,
                    want_llr=want_llr, want_hard=want_hard, want_fft=want_fft,
                    ref_hard=(dev(bits) if want_cnt else None), counters=counters)
    if want_llr:
        assert same(out.llr.cpu().numpy(), ref["llr"])
    else:
        assert out.llr is None
    if want_hard:
        assert same(out.hard.cpu().numpy(), ref["hard"])
    if want_fft:
        assert same(out.fft.cpu().numpy(), ref["fft"])
    if want_cnt:
        c = counters.cpu().numpy()
        assert c[0] == oracle.count_bit_errors(ref["hard"], bits)
        assert c[1] == n_sym * n * bps and c[3] == n_sym * n and c[4] == n_sym
        l = ref["llr"].reshape(-1, bps)
        q = oracle.llr_abs_units(l)
        assert c[5] == int(q.sum())
        assert c[6] == int(((l < 0).astype(np.uint8) != np.unpackbits(bits).reshape(-1, bps)).sum())


@pytest.mark.parametrize("n", [64, 128, 256, 512, 1024, 2048, 4096, 8192])
@pytest.mark.parametrize("bps", [2, 4, 6, 8, 10])
def test_demod_llr_only_every_shape(ctx, oracle, n, bps):
    """The headline instantiation (LLRs as the only output, no taps) for every tuned size and order,
    with a symbol count that leaves the last wave of the small sizes partly empty."""
    from sdr_amd.link import noise_sigma, llr_weight
    cp = n // 4
    n_sym = 13 if n <= 2048 else 5
    ebn0 = {2: 5.0, 4: 9.0, 6: 14.0, 8: 19.0, 10: 24.0}[bps]
    rx, _ = oracle.link_generate(n, cp, bps, 17 * n + bps, 2, n_sym, np.float32(noise_sigma(n, bps, ebn0)))
    w = np.float32(llr_weight(bps, ebn0))
    ref = oracle.demod(rx, n, cp, bps, weight=w, want_fft=False, want_hard=False)
    guard = torch.full((n_sym * n * bps + 64,), 7.5, dtype=torch.float32, device="cuda")      # nothing may land behind the last symbol
    out = ctx.demod(dev(rx), n, cp, bps, llr_weight=float(w), out_llr=guard[: n_sym * n * bps].view(n_sym, n, bps))
    assert same(out.llr.cpu().numpy(), ref["llr"])
    assert bool((guard[n_sym * n * bps:] == 7.5).all())


@pytest.mark.parametrize("n,cp,bps,per", [(64, 16, 2, 7), (256, 64, 4, 7), (1024, 256, 4, 7), (1024, 256, 6, 7), (2048, 512, 6, 7), (4096, 1024, 8, 7), (512, 128, 2, 7),
                                          (128, 32, 6, 7), (256, 64, 8, 7), (64, 16, 6, 7),
                                          # runs of seven symbols per workgroup, sets aligned with the runs and not
                                          (4096, 1024, 2, 7), (4096, 1024, 1, 5), (2048, 512, 10, 14), (2048, 512, 2, 3), (4096, 1024, 10, 5), (2048, 512, 1, 14),
                                          # taps requested four register groups ahead (N = 8192)
                                          (8192, 2048, 1, 7), (8192, 2048, 2, 5), (8192, 2048, 8, 14), (1024, 256, 10, 14)])
@pytest.mark.parametrize("which", ["both", "taps", "weights"])
def test_demod_llr_only_with_taps_or_weights(ctx, oracle, n, cp, bps, per, which):
    """LLR-only output takes the tap-prefetching instantiation; either array may be absent."""
    from sdr_amd.link import noise_sigma
    rng = np.random.default_rng(n + bps)
    n_sym = 31 if n <= 4096 else 17
    n_sets = (n_sym + per - 1) // per
    h = (rand_c(rng, (n_sets, n)) * np.float32(np.sqrt(0.5))).astype(np.complex64)
    h[np.abs(h) < 0.1] += np.float32(0.3)
    rx, _ = oracle.link_generate(n, cp, bps, 7, 0, n_sym, np.float32(noise_sigma(n, bps, 25.0)), chan=h, sym_per_set=per)
    taps, wgt = oracle.zf_taps(bps, h, 20.0)
    eq = taps if which != "weights" else None
    wt = wgt if which != "taps" else np.float32(1.75)
    ref = oracle.demod(rx, n, cp, bps, eq=eq, weight=wt, sym_per_set=per, want_fft=False, want_hard=False)
    out = ctx.demod(dev(rx), n, cp, bps, eq=(dev(eq.ravel()) if eq is not None else None),
                    llr_weight=(dev(wt.ravel()) if which != "taps" else float(wt)), sym_per_set=per)
    assert rel_err(out.llr.cpu().numpy(), ref["llr"]) <= TOL
    assert same(out.llr.cpu().numpy(), ref["llr"])


@pytest.mark.parametrize("n,cp,bps,per", [(64, 16, 2, 7), (128, 32, 6, 3), (256, 64, 8, 14), (512, 128, 4, 5), (1024, 256, 6, 14), (1024, 256, 4, 14), (1024, 256, 10, 2),
                                          (2048, 512, 6, 14), (2048, 512, 2, 7), (4096, 1024, 8, 14), (4096, 1024, 1, 7), (8192, 2048, 6, 14), (8192, 2048, 1, 5),
                                          (1024, 255, 6, 14), (16384, 4096, 4, 2), (32, 8, 4, 3)])
def test_demod_weight_from_taps(ctx, oracle, n, cp, bps, per):
    """SDR_DEMOD_WEIGHT_FROM_TAPS: taps without a weight array, w_k = llr_weight0 / |G_k|^2 formed in the kernel (one
    correctly rounded division per bin).  The LLR-only request takes the instantiation that issues no weight load at all,
    the every-output request the ordinary one, an odd prefix or an untuned size the generic engines: all equal to the
    oracle's definition bit for bit, and within 1e-6 of the two-array form (zero-forcing weights norm^2 |H|^2 / N0)."""
    from sdr_amd.link import noise_sigma, grid_energy
    rng = np.random.default_rng(3 * n + bps)
    n_sym = 31 if n <= 2048 else 9
    n_sets = (n_sym + per - 1) // per
    h = (rand_c(rng, (n_sets, n)) * np.float32(np.sqrt(0.5))).astype(np.complex64)
    h[np.abs(h) < 0.1] += np.float32(0.3)
    rx, bits = oracle.link_generate(n, cp, bps, 17, 0, n_sym, np.float32(noise_sigma(n, bps, 25.0)), chan=h, sym_per_set=per)
    inv_n0 = 20.0
    taps, wgt = oracle.zf_taps(bps, h, inv_n0)
    w0 = np.float32(np.float32(1.0 / grid_energy(bps)) * np.float32(inv_n0))
    ref = oracle.demod(rx, n, cp, bps, eq=taps, weight=w0, sym_per_set=per, weight_from_taps=True)
    d_rx, d_taps = dev(rx), dev(taps.ravel())
    out = ctx.demod(d_rx, n, cp, bps, eq=d_taps, llr_weight=float(w0), sym_per_set=per, weight_from_taps=True)
    assert same(out.llr.cpu().numpy(), ref["llr"])
    two = oracle.demod(rx, n, cp, bps, eq=taps, weight=wgt, sym_per_set=per, want_fft=False, want_hard=False)
    assert rel_err(ref["llr"], two["llr"]) < 2e-6                       # the same weights up to rounding
    cnt = torch.zeros(8, dtype=torch.int64, device="cuda")
    full = ctx.demod(d_rx, n, cp, bps, eq=d_taps, llr_weight=float(w0), sym_per_set=per, weight_from_taps=True, want_hard=True, want_fft=True,
                     ref_hard=dev(bits), counters=cnt)
    assert same(full.llr.cpu().numpy(), ref["llr"]) and same(full.hard.cpu().numpy(), ref["hard"]) and same(full.fft.cpu().numpy(), ref["fft"])
    l = ref["llr"].reshape(-1, bps)
    q = oracle.llr_abs_units(l)
    c = cnt.cpu().numpy()
    assert c[0] == oracle.count_bit_errors(ref["hard"], bits) and c[5] == int(q.sum())
    # taps alone WITHOUT the flag: the scalar weight (and no weight load either)
    plain = oracle.demod(rx, n, cp, bps, eq=taps, weight=w0, sym_per_set=per, want_fft=False, want_hard=False)
    assert same(ctx.demod(d_rx, n, cp, bps, eq=d_taps, llr_weight=float(w0), sym_per_set=per).llr.cpu().numpy(), plain["llr"])


def test_demod_args_flags_and_round2_layout(ctx, oracle):
    """The 96-byte round-2 sdr_demod_args (no `flags`) is still accepted; the flag needs taps and no weight array."""
    import ctypes
    from sdr_amd import capi
    rx, _ = oracle.link_generate(64, 16, 4, 3, 0, 4, 0.01)
    ref = oracle.demod(rx, 64, 16, 4, weight=2.0, want_fft=False, want_hard=False)
    d_rx = dev(rx)
    llr = torch.zeros(4 * 64 * 4, dtype=torch.float32, device="cuda")
    a = capi.DemodArgs()
    a.struct_size = 96
    a.n_fft, a.cp, a.mod, a.n_sym, a.llr_weight0 = 64, 16, 4, 4, 2.0
    a.d_rx, a.d_llr = d_rx.data_ptr(), llr.data_ptr()
    a.flags = 0xFFFF                                                      # behind the declared size: must not be read
    assert ctx.lib.sdr_ofdm_demod_c32(ctx._h, ctypes.byref(a)) == capi.SDR_OK
    torch.cuda.synchronize()
    assert same(llr.cpu().numpy().reshape(ref["llr"].shape), ref["llr"])
    a.struct_size = ctypes.sizeof(capi.DemodArgs)
    assert ctx.lib.sdr_ofdm_demod_c32(ctx._h, ctypes.byref(a)) == capi.SDR_ERR_INVALID and "flags" in capi.last_error()
    a.flags = capi.DEMOD_WEIGHT_FROM_TAPS                                 # no taps given
    assert ctx.lib.sdr_ofdm_demod_c32(ctx._h, ctypes.byref(a)) == capi.SDR_ERR_INVALID and "d_eq" in capi.last_error()
    a.struct_size = 100
    assert ctx.lib.sdr_ofdm_demod_c32(ctx._h, ctypes.byref(a)) == capi.SDR_ERR_INVALID and "struct_size" in capi.last_error()


@pytest.mark.parametrize("n,cp,bps", [(64, 16, 2), (128, 32, 6), (512, 128, 8), (1024, 256, 4), (1024, 256, 6), (2048, 512, 6), (4096, 1024, 8),
                                      (8192, 2048, 4), (1024, 256, 10), (256, 64, 1)])
def test_hard_decisions_only(ctx, oracle, n, cp, bps):
    """The reference's own chain, ofdm::rx -> from_constl (ofdm.hpp:37-43, modulation.hpp:96-112),
    as one launch with packed hard bits as the only output (an instantiation without LLR arithmetic)."""
    from sdr_amd.link import noise_sigma
    n_sym = 21 if n <= 1024 else 7
    ebn0 = {1: 4.0, 2: 5.0, 4: 9.0, 6: 14.0, 8: 19.0, 10: 24.0}[bps]
    rx, bits = oracle.link_generate(n, cp, bps, 31 + n, 3, n_sym, np.float32(noise_sigma(n, bps, ebn0)))
    ref = oracle.demod(rx, n, cp, bps, want_fft=False, want_llr=False)
    out = ctx.demod(dev(rx), n, cp, bps, want_llr=False, want_hard=True)
    assert out.llr is None and same(out.hard.cpu().numpy(), ref["hard"])
    assert 0 < oracle.count_bit_errors(ref["hard"], bits) < 0.2 * bits.size * 8
    # weights cannot change a hard decision
    w = torch.rand(n, dtype=torch.float32, device="cuda") + 0.5
    again = ctx.demod(dev(rx), n, cp, bps, llr_weight=w, want_llr=False, want_hard=True)
    assert torch.equal(again.hard, out.hard)


@pytest.mark.parametrize("n,cp,bps", [(64, 16, 4), (128, 32, 6), (256, 64, 8), (1024, 256, 4), (1024, 256, 6), (1024, 256, 8), (2048, 512, 6), (4096, 1024, 8),
                                      (8192, 2048, 6), (1024, 256, 10), (512, 128, 2), (64, 16, 2), (128, 32, 2), (1024, 256, 2), (2048, 512, 2), (4096, 1024, 2),
                                      (256, 64, 1), (1024, 256, 1), (2048, 512, 1)])
def test_hard_decisions_on_boundaries_through_the_fused_kernels(ctx, oracle, n, cp, bps):
    """Bins that land within an ulp or two of decision boundaries, levels, the grid's edge and far outside it: every
    subcarrier carries an integer grid value (even = boundary, odd = level) or an outlier, the symbol goes through the
    oracle's ofdm::tx, and the device transform brings it back to within rounding of where it started.  Half the points
    therefore fail the slicer's margin test and take the kernels' exact path -- in the hard-only kernel the deferred
    fix-up of doubtful lanes (rx_fused.hpp), in the every-output and counting kernels the per-point branch -- and must
    still agree with the reference's scan (modulation.hpp:49-67) bit for bit."""
    rng = np.random.default_rng(900 + n + bps + 100000 * int(os.environ.get("SDR_SOAK_SEED", 0)))
    mi = (bps + 1) // 2
    levels = 1 << mi
    lq = 1 << (bps // 2)                                                     # BPSK: one level on Q
    inorm = np.sqrt(np.float32((levels * levels - 1) / 3.0 + (lq * lq - 1) / 3.0))
    n_sym = 9 if n <= 1024 else 3
    grid = rng.integers(-levels - 3, levels + 4, size=(n_sym, n, 2)).astype(np.float64)
    far = rng.random((n_sym, n, 2)) < 0.01                                   # outliers around the slicer's reach limit
    grid = np.where(far, rng.choice([-1.0, 1.0], size=grid.shape) * (levels - 1 + rng.choice([30.0, 58.0, 61.0, 62.0, 63.0, 70.0], size=grid.shape)), grid)
    bins = ((grid[..., 0] + 1j * grid[..., 1]) / float(inorm)).astype(np.complex64)
    rc, rx = oracle.ofdm_tx(bins, n, cp)                                      # (n_sym, n + cp)
    assert rc == 0
    ref = oracle.demod(rx, n, cp, bps, weight=1.5)
    hard_only = ctx.demod(dev(rx), n, cp, bps, want_llr=False, want_hard=True)
    assert same(hard_only.hard.cpu().numpy(), ref["hard"])
    both = ctx.demod(dev(rx), n, cp, bps, llr_weight=1.5, want_llr=True, want_hard=True)
    assert same(both.hard.cpu().numpy(), ref["hard"]) and same(both.llr.cpu().numpy(), ref["llr"])
    # and the counting kernel against reference bits that differ from the decisions in known places
    refbits = ref["hard"].copy()
    refbits[::7] ^= 0x5A
    cnt = torch.zeros(8, dtype=torch.int64, device="cuda")
    ctx.demod(dev(rx), n, cp, bps, llr_weight=1.5, want_llr=True, ref_hard=dev(refbits), counters=cnt)
    assert int(cnt[0]) == oracle.count_bit_errors(ref["hard"], refbits)
    # the points really are where the margin test gives up: a tenth of them at least lie within 2^-12 of a boundary
    u = ref["fft"].astype(np.complex128) * float(inorm)
    dist = np.minimum(np.abs(np.mod(u.real, 2.0)), np.abs(np.mod(u.imag, 2.0)))
    dist = np.minimum(dist, 2.0 - dist)
    assert np.mean(dist < 2.0 ** -12) > 0.1


def test_demod_outputs_are_optional(ctx, oracle):
    rx, bits = oracle.link_generate(1024, 256, 4, 3, 0, 4, 0.01)
    ref = oracle.demod(rx, 1024, 256, 4, weight=2.0)
    a = ctx.demod(dev(rx), 1024, 256, 4, llr_weight=2.0, want_llr=False, want_hard=True)
    assert a.llr is None and same(a.hard.cpu().numpy(), ref["hard"])
    b = ctx.demod(dev(rx), 1024, 256, 4, llr_weight=2.0, want_llr=True, want_hard=False)
    assert b.hard is None and same(b.llr.cpu().numpy(), ref["llr"])
    c = ctx.demod(dev(rx), 1024, 256, 4, want_llr=False, want_fft=True)
    assert same(c.fft.cpu().numpy(), ref["fft"])


# ---- generator / AWGN / Monte-Carlo link ---------------------------------------------------------
@pytest.mark.parametrize("n,cp,bps", [(64, 16, 2), (1024, 256, 4), (2048, 512, 6), (4096, 1024, 8), (256, 64, 1), (512, 128, 10), (128, 32, 8), (512, 128, 6),
                                      (32, 8, 4), (8192, 2048, 2), (1024, 1024, 4), (1024, 0, 2)])
def test_link_generate_bit_exact(ctx, oracle, n, cp, bps):
    n_sym = 6
    rx, bits = ctx.link_generate(n, cp, bps, seed=7, first_symbol=3, n_sym=n_sym, sigma=0.02)
    rrx, rbits = oracle.link_generate(n, cp, bps, 7, 3, n_sym, 0.02)
    assert same(bits.cpu().numpy(), rbits)
    assert same(rx.cpu().numpy(), rrx), rel_err(rx.cpu().numpy(), rrx)


def test_link_generate_with_channel(ctx, oracle):
    rng = np.random.default_rng(1)
    n, cp, bps, per = 256, 64, 4, 14
    h = rand_c(rng, (3, n))
    rx, bits = ctx.link_generate(n, cp, bps, 5, 14, 30, 0.01, chan=dev(h), sym_per_set=per)
    rrx, rbits = oracle.link_generate(n, cp, bps, 5, 14, 30, 0.01, chan=h, sym_per_set=per)
    assert same(bits.cpu().numpy(), rbits) and same(rx.cpu().numpy(), rrx)


@pytest.mark.parametrize("n,cp,bps,ebn0", [(64, 16, 2, 4.0), (1024, 256, 4, 8.0), (1024, 256, 6, 12.0), (1024, 256, 8, 16.0),
                                           (2048, 512, 6, 12.0), (4096, 1024, 8, 18.0), (256, 64, 2, 3.0), (128, 32, 4, 7.0), (512, 128, 1, 3.0),
                                           (32, 8, 2, 4.0), (8192, 2048, 4, 8.0), (1024, 256, 10, 20.0)])
def test_link_ber_counters_bit_exact(ctx, oracle, n, cp, bps, ebn0):
    from sdr_amd.link import noise_sigma, llr_weight
    n_sym = 28 if n <= 1024 else 14
    sigma = np.float32(noise_sigma(n, bps, ebn0))
    w = np.float32(llr_weight(bps, ebn0))
    got = ctx.link_ber(n, cp, bps, seed=21, first_symbol=42, n_sym=n_sym, sigma=float(sigma), llr_weight=float(w)).cpu().numpy()
    ref = oracle.link_ber(n, cp, bps, 21, 42, n_sym, sigma, w)
    assert same(got.astype(np.uint64), ref), (got, ref)
    assert got[0] > 0


def test_link_ber_is_split_invariant_and_accumulates(ctx):
    from sdr_amd.link import noise_sigma, llr_weight
    n, cp, bps = 1024, 256, 4
    sigma, w = noise_sigma(n, bps, 8.0), llr_weight(bps, 8.0)
    whole = ctx.link_ber(n, cp, bps, 5, 0, 56, sigma, w).cpu().numpy()
    acc = torch.zeros(8, dtype=torch.int64, device="cuda")
    for first, cnt in ((0, 14), (14, 28), (42, 14)):
        ctx.link_ber(n, cp, bps, 5, first, cnt, sigma, w, counters=acc)
    assert same(acc.cpu().numpy(), whole)


def test_large_batch_properties(ctx):
    """Full-size behaviour through size-independent properties: tx -> rx round trip and
    linearity of the batched FFT on a batch far larger than the oracle could check."""
    n, cp, s = 1024, 256, 16384
    g = torch.Generator(device="cuda").manual_seed(3)
    a = torch.randn((s, n), dtype=torch.float32, device="cuda", generator=g).to(torch.complex64)
    a = a + 1j * torch.randn((s, n), dtype=torch.float32, device="cuda", generator=g)
    back = ctx.ofdm_rx(ctx.ofdm_tx(a, n, cp), n, cp)
    assert ((back - a).abs().amax(dim=1) / a.abs().amax(dim=1)).max().item() < 1e-5
    b = torch.roll(a, 1, dims=0)
    fa = ctx.ofdm_rx(torch.cat([torch.zeros((s, cp), dtype=a.dtype, device="cuda"), a], dim=1), n, cp)
    fb = ctx.ofdm_rx(torch.cat([torch.zeros((s, cp), dtype=a.dtype, device="cuda"), b], dim=1), n, cp)
    fab = ctx.ofdm_rx(torch.cat([torch.zeros((s, cp), dtype=a.dtype, device="cuda"), a + b], dim=1), n, cp)
    assert ((fab - fa - fb).abs().amax(dim=1) / fab.abs().amax(dim=1)).max().item() < 1e-5
    ref = torch.fft.fft(a[:64].to(torch.complex128), dim=1)
    assert ((fa[:64] - ref).abs().amax(dim=1) / ref.abs().amax(dim=1)).max().item() < 5e-6


@pytest.mark.parametrize("n,cp,bps,ebn0", [(1024, 256, 4, 9.0), (1024, 256, 6, 13.0), (2048, 512, 6, 13.0), (4096, 1024, 8, 19.0), (64, 16, 2, 5.0)])
def test_full_size_two_paths_agree(ctx, n, cp, bps, ebn0):
    """Size-independent property at a batch far beyond what the oracle can check: the frame
    generator + fused receive kernel (through HBM) and the fully fused Monte-Carlo kernel (nothing
    in HBM) are different code paths over the same random streams, so all their counters must be
    identical; and hard bits with LLR signs must reproduce the error count."""
    from sdr_amd.link import noise_sigma, llr_weight
    n_sym = (1 << 26) // n                      # 64 Mi subcarriers per case
    sigma, w = noise_sigma(n, bps, ebn0), llr_weight(bps, ebn0)
    rx, bits = ctx.link_generate(n, cp, bps, seed=77, first_symbol=1000, n_sym=n_sym, sigma=sigma)
    c1 = torch.zeros(8, dtype=torch.int64, device="cuda")
    out = ctx.demod(rx, n, cp, bps, llr_weight=w, want_hard=True, ref_hard=bits, counters=c1)
    c2 = ctx.link_ber(n, cp, bps, seed=77, first_symbol=1000, n_sym=n_sym, sigma=sigma, llr_weight=w)
    assert torch.equal(c1, c2), (c1.tolist(), c2.tolist())
    assert c1[1].item() == n_sym * n * bps and c1[0].item() > 0
    diff = torch.bitwise_xor(out.hard, bits)
    lut = torch.tensor([bin(i).count("1") for i in range(256)], dtype=torch.int64, device="cuda")
    assert int(lut[diff.long()].sum().item()) == c1[0].item()
    # LLR-only variant of the kernel writes the same LLRs as the all-outputs variant
    llr_only = ctx.demod(rx, n, cp, bps, llr_weight=w).llr
    assert torch.equal(llr_only, out.llr)
    sign_bits = (llr_only < 0).reshape(-1, 8).to(torch.uint8)
    packed = (sign_bits * torch.tensor([128, 64, 32, 16, 8, 4, 2, 1], dtype=torch.uint8, device="cuda")).sum(dim=1).to(torch.uint8)
    assert int(lut[torch.bitwise_xor(packed, bits).long()].sum().item()) == c1[6].item()


def test_full_size_equalised_round_trip(ctx):
    """BASELINE configs[2] shape (FFT-2048, 64-QAM, one channel set per 14-symbol frame) at one
    resident chunk: generate through a random channel -> ZF taps -> fused receive.  Properties
    that need no oracle: at high SNR nearly every bit comes back; the kernel's own counters equal
    a recount of its hard bits; the tap-prefetching LLR-only instantiation and the all-outputs
    instantiation write identical LLRs; splitting the chunk at a frame boundary changes nothing."""
    from sdr_amd.link import noise_sigma
    n, cp, bps, per = 2048, 512, 6, 14
    n_sym = per * 2340                                   # 32760 symbols, whole frames
    g = torch.Generator(device="cuda").manual_seed(11)
    n_sets = n_sym // per
    h = torch.complex(torch.randn((n_sets, n), device="cuda", generator=g), torch.randn((n_sets, n), device="cuda", generator=g)) * (0.5 ** 0.5)
    h = torch.where(h.abs() < 0.25, h + 0.5, h).to(torch.complex64).contiguous()     # keep |H| away from 0
    ebn0 = 45.0
    rx, bits = ctx.link_generate(n, cp, bps, seed=3, first_symbol=0, n_sym=n_sym, sigma=noise_sigma(n, bps, ebn0), chan=h, sym_per_set=per)
    taps, wgt = ctx.zf_taps(h.view(-1), bps, bps * 10 ** (ebn0 / 10))
    cnt = torch.zeros(8, dtype=torch.int64, device="cuda")
    full = ctx.demod(rx, n, cp, bps, eq=taps, llr_weight=wgt, sym_per_set=per, want_hard=True, ref_hard=bits, counters=cnt)
    c = cnt.tolist()
    assert c[1] == n_sym * n * bps and c[4] == n_sym
    assert c[0] < 1e-4 * c[1]                                            # ZF at 45 dB: a handful of deep-fade errors at most
    lut = torch.tensor([bin(i).count("1") for i in range(256)], dtype=torch.int64, device="cuda")
    assert int(lut[torch.bitwise_xor(full.hard, bits).long()].sum().item()) == c[0]
    llr_only = ctx.demod(rx, n, cp, bps, eq=taps, llr_weight=wgt, sym_per_set=per).llr
    assert torch.equal(llr_only, full.llr)
    cut = per * 1000
    a = ctx.demod(rx[:cut], n, cp, bps, eq=taps, llr_weight=wgt, sym_per_set=per).llr
    b = ctx.demod(rx[cut:], n, cp, bps, eq=taps.view(n_sets, n)[cut // per:], llr_weight=wgt.view(n_sets, n)[cut // per:], sym_per_set=per).llr
    assert torch.equal(torch.cat([a, b]), full.llr)


# ---- streaming ingest ----------------------------------------------------------------------------
@pytest.mark.parametrize("n,cp,bps,chunk,depth", [(1024, 256, 4, 64, 3), (64, 16, 2, 100, 2), (2048, 512, 6, 16, 4)])
def test_pipe_equals_one_shot_demod(ctx, oracle, n, cp, bps, chunk, depth):
    """Chunks through the three-stream ring give exactly what one call on the whole array gives
    (and what the oracle gives), including a ragged last chunk and more chunks than slots."""
    from sdr_amd.link import noise_sigma
    from sdr_amd.pipe import Pipe
    n_sym = chunk * (2 * depth + 1) + chunk // 2
    rx, _ = oracle.link_generate(n, cp, bps, 77, 0, n_sym, np.float32(noise_sigma(n, bps, 10.0)))
    rx = rx.reshape(n_sym, n + cp)
    ref = oracle.demod(rx, n, cp, bps, weight=1.25, want_fft=False)
    with Pipe(0, n_fft=n, cp=cp, mod=bps, chunk_symbols=chunk, depth=depth, llr_to_host=True, hard_to_host=True, llr_weight=1.25) as pipe:
        llr, hard, seqs = [], [], []
        for c in pipe.run(rx[i:i + chunk] for i in range(0, n_sym, chunk)):
            llr.append(c.llr.copy()); hard.append(c.hard.copy()); seqs.append(c.sequence)
            assert c.d_llr and c.d_hard
        assert seqs == list(range(len(seqs))) and pipe.in_flight() == 0
    assert same(np.concatenate(llr), ref["llr"])
    assert same(np.concatenate(hard), ref["hard"])


def test_pipe_with_taps_and_weights(ctx, oracle):
    """The pipe's equaliser sets restart at every chunk (set 0 = the chunk's first symbol)."""
    from sdr_amd.link import noise_sigma
    from sdr_amd.pipe import Pipe
    n, cp, bps, per, chunk = 256, 64, 6, 4, 12
    rng = np.random.default_rng(8)
    n_sets = chunk // per
    h = (rand_c(rng, (n_sets, n)) * np.float32(np.sqrt(0.5))).astype(np.complex64)
    h[np.abs(h) < 0.2] += np.float32(0.4)
    taps, wgt = oracle.zf_taps(bps, h, 30.0)
    d_taps, d_wgt = dev(taps.ravel()), dev(wgt.ravel())
    n_chunks = 5
    rx = [oracle.link_generate(n, cp, bps, 5, c * chunk, chunk, np.float32(noise_sigma(n, bps, 30.0)), chan=h, sym_per_set=per)[0].reshape(chunk, n + cp)
          for c in range(n_chunks)]
    with Pipe(0, n_fft=n, cp=cp, mod=bps, chunk_symbols=chunk, depth=2, llr_to_host=True, hard_to_host=True,
              sym_per_set=per, d_eq=d_taps.data_ptr(), d_llr_weight=d_wgt.data_ptr()) as pipe:
        got = [(c.llr.copy(), c.hard.copy()) for c in pipe.run(rx)]
    assert len(got) == n_chunks
    for x, (llr, hard) in zip(rx, got):
        ref = oracle.demod(x, n, cp, bps, eq=taps, weight=wgt, sym_per_set=per, want_fft=False)
        assert same(llr, ref["llr"]) and same(hard, ref["hard"])


def test_pipe_protocol_errors(ctx):
    from sdr_amd import capi
    from sdr_amd.pipe import Pipe
    with pytest.raises(RuntimeError):
        Pipe(0, n_fft=1000, cp=0, mod=4, chunk_symbols=8, hard_to_host=True)          # not a power of two
    assert capi.last_error() == "The sequence size must be of powers of 2"
    with pytest.raises(RuntimeError):
        Pipe(0, n_fft=64, cp=16, mod=4, chunk_symbols=8)                               # no output asked for
    with Pipe(0, n_fft=64, cp=16, mod=4, chunk_symbols=8, depth=2, hard_to_host=True) as pipe:
        with pytest.raises(RuntimeError):
            pipe.submit(4)                                                              # submit without acquire
        with pytest.raises(RuntimeError):
            pipe.collect()                                                              # nothing submitted
        for _ in range(2):
            pipe.acquire()[:] = 0
            pipe.submit(8)
        with pytest.raises(RuntimeError):
            pipe.acquire()                                                              # ring full
        with pytest.raises(RuntimeError):
            pipe.release()                                                              # nothing collected
        c = pipe.collect()
        assert c.n_sym == 8 and c.sequence == 0
        pipe.release()
        buf = pipe.acquire()
        with pytest.raises(RuntimeError):
            pipe.submit(9)                                                              # more than a slot holds
        pipe.submit(1)
        assert pipe.in_flight() == 2


# ---- argument errors surface as values, never as faults --------------------------------------------
def test_c_abi_argument_errors(ctx):
    import ctypes
    from sdr_amd import capi
    lib, h = ctx.lib, ctx._h
    rx = torch.zeros(4 * 80, dtype=torch.complex64, device="cuda")
    llr = torch.zeros(4 * 64 * 4, dtype=torch.float32, device="cuda")

    def args(**kw):
        a = capi.DemodArgs()
        a.struct_size = ctypes.sizeof(capi.DemodArgs)
        a.n_fft, a.cp, a.mod, a.n_sym = 64, 16, 4, 4
        a.d_rx, a.d_llr = rx.data_ptr(), llr.data_ptr()
        for k, v in kw.items():
            setattr(a, k, v)
        return a

    assert lib.sdr_ofdm_demod_c32(h, ctypes.byref(args())) == capi.SDR_OK
    assert lib.sdr_ofdm_demod_c32(h, None) == capi.SDR_ERR_INVALID
    assert lib.sdr_ofdm_demod_c32(h, ctypes.byref(args(struct_size=8))) == capi.SDR_ERR_INVALID
    assert "struct_size" in capi.last_error()
    assert lib.sdr_ofdm_demod_c32(h, ctypes.byref(args(mod=3))) == capi.SDR_ERR_INVALID
    assert lib.sdr_ofdm_demod_c32(h, ctypes.byref(args(n_fft=48))) == capi.SDR_ERR_NOT_POW2
    assert capi.last_error() == "The sequence size must be of powers of 2"
    assert lib.sdr_ofdm_demod_c32(h, ctypes.byref(args(d_rx=None))) == capi.SDR_ERR_INVALID
    cnt = torch.zeros(8, dtype=torch.int64, device="cuda")
    assert lib.sdr_ofdm_demod_c32(h, ctypes.byref(args(d_counters=cnt.data_ptr()))) == capi.SDR_ERR_INVALID   # counters without reference bits
    assert lib.sdr_ofdm_demod_c32(h, ctypes.byref(args(n_sym=0))) == capi.SDR_OK                                # empty batch: nothing to do
    torch.cuda.synchronize()
    # host form: too few tap / weight sets for what the kernel would index is refused, not read out of bounds
    h_rx = np.zeros(30 * 80, dtype=np.complex64)
    h_llr = np.zeros(30 * 64 * 4, dtype=np.float32)
    h_eq = np.ones(2 * 64, dtype=np.complex64)
    h_w = np.ones(2 * 64, dtype=np.float32)
    hp = lambda a_: a_.ctypes.data_as(ctypes.c_void_p).value
    host = lambda **kw: args(**{**dict(n_sym=30, sym_per_set=14, d_rx=hp(h_rx), d_llr=hp(h_llr)), **kw})
    assert lib.sdr_ofdm_demod_host_c32(h, ctypes.byref(host(d_eq=hp(h_eq))), 2) == capi.SDR_ERR_INVALID       # 30 symbols / 14 = 3 sets
    assert "n_sets" in capi.last_error()
    assert lib.sdr_ofdm_demod_host_c32(h, ctypes.byref(host(d_llr_weight=hp(h_w))), 0) == capi.SDR_ERR_INVALID
    assert lib.sdr_ofdm_demod_host_c32(h, ctypes.byref(host(d_eq=hp(h_eq), sym_per_set=0)), 0) == capi.SDR_ERR_INVALID   # one set needed
    assert lib.sdr_ofdm_demod_host_c32(h, ctypes.byref(host(d_eq=hp(h_eq), sym_per_set=15)), 2) == capi.SDR_OK
    assert lib.sdr_ofdm_demod_host_c32(h, ctypes.byref(host()), 0) == capi.SDR_OK                              # no taps: n_sets ignored
    with pytest.raises(ValueError):
        ctx.demod(rx, 64, 16, 4, eq=torch.ones(64, dtype=torch.complex64, device="cuda"), sym_per_set=2)        # 4 symbols / 2 = 2 sets needed
    with pytest.raises(ValueError):
        ctx.demod(rx, 64, 16, 4, llr_weight=torch.ones(63, dtype=torch.float32, device="cuda"))


def test_two_contexts_in_two_threads(oracle):
    """Contexts are independent (include/sdr_amd.h): two host threads, each with a context and a
    stream of its own, run the counting receive path concurrently and both match the oracle."""
    import threading
    from sdr_amd.ops import Context
    from sdr_amd.link import noise_sigma
    n, cp, bps, n_sym = 1024, 256, 4, 64
    rx, bits = oracle.link_generate(n, cp, bps, 21, 0, n_sym, np.float32(noise_sigma(n, bps, 8.0)))
    ref = oracle.demod(rx, n, cp, bps, weight=1.0, want_fft=False)
    want_errors = oracle.count_bit_errors(ref["hard"], bits)
    results, failures = {}, []

    def worker(tag):
        try:
            stream = torch.cuda.Stream()
            with torch.cuda.stream(stream):
                c = Context(0)
                drx, dbits = dev(rx), dev(bits)
                cnt = torch.zeros(8, dtype=torch.int64, device="cuda")
                for _ in range(20):
                    out = c.demod(drx, n, cp, bps, llr_weight=1.0, want_hard=True, ref_hard=dbits, counters=cnt)
                stream.synchronize()
                results[tag] = (out.llr.cpu().numpy(), out.hard.cpu().numpy(), cnt.cpu().numpy())
                c.close()
        except Exception as e:      # surfaced below: an assertion in a thread would be lost
            failures.append((tag, repr(e)))

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not failures, failures
    for tag in range(2):
        llr, hard, cnt = results[tag]
        assert same(llr, ref["llr"]) and same(hard, ref["hard"])
        assert cnt[0] == 20 * want_errors and cnt[4] == 20 * n_sym


def test_copy_shape_diagnostic_touches_every_output(ctx):
    n, cp, bps, n_sym = 1024, 256, 6, 37
    rx = torch.ones((n_sym, n + cp), dtype=torch.complex64, device="cuda")
    out = torch.full((n_sym, n, bps), float("nan"), dtype=torch.float32, device="cuda")
    ctx.copy_shape(rx, n, cp, bps, out)
    torch.cuda.synchronize()
    assert not torch.isnan(out).any()


def test_contexts_and_pipes_release_their_memory(oracle):
    """Create / use / destroy cycles leave device memory where it was (twiddle tables, staging,
    counter slots, pipe slots and pinned buffers are all owned by the handle)."""
    from sdr_amd.ops import Context
    from sdr_amd.pipe import Pipe
    rx, bits = oracle.link_generate(256, 64, 4, 1, 0, 8, np.float32(0.05))
    drx, dbits = dev(rx), dev(bits)

    def cycle():
        c = Context(0)
        cnt = torch.zeros(8, dtype=torch.int64, device="cuda")
        c.demod(drx, 256, 64, 4, want_hard=True, want_fft=True, ref_hard=dbits, counters=cnt)
        c.fft2_(torch.zeros(3 * 4096, dtype=torch.complex128, device="cuda").view(3, 4096))
        c.link_ber(256, 64, 4, 1, 0, 64, 0.05, 1.0)
        c.synchronize()
        c.close()
        with Pipe(0, n_fft=256, cp=64, mod=4, chunk_symbols=8, depth=3, llr_to_host=True, hard_to_host=True) as p:
            p.acquire()[:] = rx.reshape(8, 320)
            p.submit(8)
            p.collect()
            p.release()

    for _ in range(3):
        cycle()
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    free0, _ = torch.cuda.mem_get_info()
    for _ in range(40):
        cycle()
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    free1, _ = torch.cuda.mem_get_info()
    assert free0 - free1 < (8 << 20), (free0, free1)
