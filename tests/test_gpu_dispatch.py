"""Dispatch parity: sdr_ofdm_demod_c32 leaves the tuned kernels for the generic engines
(fft_generic_f32 + demap_after_fft, sdr_amd/csrc/capi.hip) when the cyclic prefix is odd, when a
buffer is not 16-byte aligned, or when n_fft lies outside 64...8192.  A caller reaches that path by
accident at exactly the sizes the tuned kernels cover, so it is compared with the oracle THERE
(and at n_fft = 16384, which no tuned kernel takes): every output bit-equal, like the tuned path.
The chain checked is the reference's ofdm::rx (lib/inc/ofdm.hpp:37-43) -> [equalise, LLR] ->
from_constl (lib/inc/modulation.hpp:96-112)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def same(a, b):
    return bool(np.array_equal(np.asarray(a), np.asarray(b)))


def rand_c(rng, shape):
    return (rng.standard_normal(shape) + 1j * rng.standard_normal(shape)).astype(np.complex64)


def shifted(src: np.ndarray, shift: int, dtype=None) -> torch.Tensor:
    """A device copy of `src` that starts `shift` ELEMENTS into a 256-byte aligned allocation."""
    t = torch.from_numpy(np.ascontiguousarray(src)).cuda().reshape(-1)
    pool = torch.empty(t.numel() + 8, dtype=t.dtype, device="cuda")
    view = pool[shift: shift + t.numel()]
    view.copy_(t)
    return view


def shifted_out(numel: int, shift: int, dtype, fill) -> tuple:
    pool = torch.full((numel + 16,), fill, dtype=dtype, device="cuda")
    return pool, pool[shift: shift + numel]


def _run_case(ctx, oracle, n, cp, bps, n_sym, taps, shifts, seed):
    """shifts = (rx, llr, hard, fft, eq, weights, ref) in elements of each buffer's own type."""
    from sdr_amd.link import noise_sigma
    s_rx, s_llr, s_hard, s_fft, s_eq, s_w, s_ref = shifts
    rng = np.random.default_rng(seed)
    per = 3
    n_sets = (n_sym + per - 1) // per
    h = (rand_c(rng, (n_sets, n)) * np.float32(np.sqrt(0.5))).astype(np.complex64)
    h[np.abs(h) < 0.1] += np.float32(0.3)
    use_chan = taps in ("both", "taps")
    ebn0 = {1: 6.0, 2: 7.0, 4: 11.0, 6: 16.0, 8: 21.0, 10: 26.0}[bps]
    rx, bits = oracle.link_generate(n, cp, bps, 300 + seed, 0, n_sym, np.float32(noise_sigma(n, bps, ebn0)),
                                    chan=(h if use_chan else None), sym_per_set=(per if use_chan else 0))
    t, w = oracle.zf_taps(bps, h, 9.0)
    eq = t if use_chan else None
    wt = w if taps in ("both", "weights") else np.float32(0.8)
    sps = per if taps != "none" else 0
    ref = oracle.demod(rx, n, cp, bps, eq=eq, weight=wt, sym_per_set=sps)

    d_rx = shifted(rx, s_rx)
    d_eq = shifted(eq, s_eq) if eq is not None else None
    d_w = shifted(wt, s_w) if isinstance(wt, np.ndarray) else float(wt)
    d_ref = shifted(bits, s_ref)
    n_llr, n_hard, n_bins = n_sym * n * bps, n_sym * n * bps // 8, n_sym * n
    p_llr, o_llr = shifted_out(n_llr, s_llr, torch.float32, 7.5)
    p_hard, o_hard = shifted_out(n_hard, s_hard, torch.uint8, 0xA5)
    p_fft, o_fft = shifted_out(n_bins, s_fft, torch.complex64, 3.25)
    cnt = torch.zeros(8, dtype=torch.int64, device="cuda")
    ctx.demod(d_rx, n, cp, bps, eq=d_eq, llr_weight=d_w, sym_per_set=sps, out_llr=o_llr.view(n_sym, n, bps), out_hard=o_hard,
              out_fft=o_fft.view(n_sym, n), ref_hard=d_ref, counters=cnt)
    torch.cuda.synchronize()
    assert same(o_fft.cpu().numpy().reshape(ref["fft"].shape), ref["fft"])
    assert same(o_llr.cpu().numpy().reshape(ref["llr"].shape), ref["llr"])
    assert same(o_hard.cpu().numpy(), ref["hard"])
    # nothing written outside the views
    assert bool((p_llr[:s_llr] == 7.5).all()) and bool((p_llr[s_llr + n_llr:] == 7.5).all())
    assert bool((p_hard[:s_hard] == 0xA5).all()) and bool((p_hard[s_hard + n_hard:] == 0xA5).all())
    assert bool((p_fft[:s_fft] == 3.25).all()) and bool((p_fft[s_fft + n_bins:] == 3.25).all())
    c = cnt.cpu().numpy()
    assert c[0] == oracle.count_bit_errors(ref["hard"], bits)
    assert c[1] == n_sym * n * bps and c[3] == n_sym * n and c[4] == n_sym
    l = ref["llr"].reshape(-1, bps)
    q = oracle.llr_abs_units(l)
    tx = np.unpackbits(bits).reshape(-1, bps)
    assert c[2] == int((tx != np.unpackbits(ref["hard"]).reshape(-1, bps)).any(axis=1).sum())
    assert c[5] == int(q.sum())
    assert c[6] == int(((l < 0).astype(np.uint8) != tx).sum())
    # the LLR-only and the hard-only requests take other instantiations on the tuned path: same answers here
    o2 = torch.empty(n_llr + 4, dtype=torch.float32, device="cuda")[s_llr: s_llr + n_llr]
    ctx.demod(d_rx, n, cp, bps, eq=d_eq, llr_weight=d_w, sym_per_set=sps, out_llr=o2.view(n_sym, n, bps))
    assert same(o2.cpu().numpy().reshape(ref["llr"].shape), ref["llr"])
    o3 = torch.empty(n_hard + 4, dtype=torch.uint8, device="cuda")[s_hard: s_hard + n_hard]
    ctx.demod(d_rx, n, cp, bps, eq=d_eq, llr_weight=d_w, sym_per_set=sps, want_llr=False, out_hard=o3)
    assert same(o3.cpu().numpy(), ref["hard"])


ALIGNED = (0, 0, 0, 0, 0, 0, 0)

# (n, cp, bps, taps, shifts): an odd cp with aligned buffers, and an even cp with ONE misaligned buffer
# each, at every tuned size; both at n = 16384 and 32 (no tuned kernel)
_ODD_CP = [(64, 5, 2, "none"), (64, 17, 10, "both"), (128, 33, 6, "weights"), (256, 63, 8, "taps"), (512, 1, 4, "both"),
           (1024, 255, 6, "both"), (1024, 3, 4, "none"), (2048, 511, 6, "both"), (4096, 1023, 8, "none"), (8192, 7, 2, "taps"),
           (1024, 1, 1, "both")]
_MISALIGNED = [
    (64, 16, 4, "both", (1, 0, 0, 0, 0, 0, 0)),       # samples 8 bytes off
    (256, 64, 6, "none", (0, 1, 0, 0, 0, 0, 0)),      # LLRs 4 bytes off
    (1024, 256, 6, "both", (0, 0, 0, 0, 1, 0, 0)),    # taps 8 bytes off
    (1024, 256, 4, "both", (0, 0, 0, 0, 0, 1, 0)),    # weights 4 bytes off
    (1024, 256, 4, "none", (0, 0, 1, 0, 0, 0, 0)),    # packed hard bits 1 byte off
    (2048, 512, 6, "weights", (0, 0, 0, 1, 0, 0, 0)), # bins 8 bytes off
    (4096, 1024, 8, "none", (0, 0, 0, 0, 0, 0, 3)),   # reference bits 3 bytes off
    (8192, 2048, 4, "both", (1, 3, 2, 1, 1, 1, 1)),   # everything off
    (512, 128, 10, "both", (1, 2, 3, 1, 1, 3, 2)),
    (128, 32, 1, "taps", (1, 1, 1, 1, 1, 0, 1)),
]


@pytest.mark.parametrize("n,cp,bps,taps", _ODD_CP)
def test_demod_odd_prefix_takes_generic_path_and_equals_oracle(ctx, oracle, n, cp, bps, taps):
    _run_case(ctx, oracle, n, cp, bps, 11 if n <= 2048 else 5, taps, ALIGNED, seed=n + cp + bps)


@pytest.mark.parametrize("n,cp,bps,taps,shifts", _MISALIGNED)
def test_demod_misaligned_buffers_take_generic_path_and_equal_oracle(ctx, oracle, n, cp, bps, taps, shifts):
    _run_case(ctx, oracle, n, cp, bps, 10 if n <= 2048 else 5, taps, shifts, seed=n + bps)


@pytest.mark.parametrize("n,cp,bps,taps,shifts", [(16384, 4096, 4, "none", ALIGNED), (16384, 4095, 6, "both", ALIGNED), (16384, 1024, 2, "weights", (1, 1, 1, 1, 1, 1, 1)),
                                                  (32, 8, 4, "both", ALIGNED), (32, 7, 6, "none", (1, 0, 0, 0, 0, 0, 0)), (16, 4, 2, "both", ALIGNED),
                                                  (32768, 16, 4, "none", ALIGNED)])
def test_demod_sizes_without_a_tuned_kernel(ctx, oracle, n, cp, bps, taps, shifts):
    """n_fft = 16384 / 32768 (stage-by-stage engine over HBM) and n_fft < 64 through sdr_ofdm_demod_c32."""
    _run_case(ctx, oracle, n, cp, bps, 3 if n >= 16384 else 9, taps, shifts, seed=n + cp)


def _random_cases(count=24, seed=20261112):
    rng = np.random.default_rng(seed)
    out = []
    for i in range(count):
        n = int(rng.choice([64, 128, 256, 512, 1024, 2048, 4096, 8192]))
        cp = int(rng.integers(0, n // 2 + 1))
        bps = int(rng.choice([1, 2, 4, 6, 8, 10]))
        taps = str(rng.choice(["none", "both", "taps", "weights"]))
        shifts = tuple(int(rng.integers(0, 4)) if j in (1, 2, 5, 6) else int(rng.integers(0, 2)) for j in range(7))
        out.append((i, n, cp, bps, taps, shifts))
    return out


@pytest.mark.parametrize("i,n,cp,bps,taps,shifts", _random_cases())
def test_demod_random_layouts(ctx, oracle, i, n, cp, bps, taps, shifts):
    """Random prefix lengths (odd ones included) and buffer offsets, as test_ofdm_tx_rx_random_layouts does for
    ofdm::rx / tx: whichever side of the dispatch a combination lands on, every output equals the oracle's."""
    _run_case(ctx, oracle, n, cp, bps, int(3 + i % 7), taps, shifts, seed=9000 + i)
