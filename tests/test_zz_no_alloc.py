"""sdr_ofdm_demod_noalloc_c32 (static device scratch, no allocation) against sdr_ofdm_demod_c32 on
the same inputs, through the C ABI: every output bit-identical, on two calls in a row (the second
smaller and at another FFT size, so the slots and the table choice are exercised twice), with the
oracle as the arbiter of the first.  Collected last (test_zz_*), passes alone and after the rest."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _case(ctx, oracle, n, cp, bps, n_sym, seed, with_eq):
    from sdr_amd.link import noise_sigma, llr_weight
    rng = np.random.default_rng(seed)
    sigma = np.float32(noise_sigma(n, bps, 12.0))
    per = 14 if with_eq else 0
    chan = eq = wgt = None
    if with_eq:
        n_sets = (n_sym + per - 1) // per
        chan = ((rng.standard_normal((n_sets, n)) + 1j * rng.standard_normal((n_sets, n))) * np.sqrt(0.5)).astype(np.complex64)
        chan[np.abs(chan) < 0.1] += np.float32(0.3)
        eq, wgt = oracle.zf_taps(bps, chan, 40.0)
    rx, bits = oracle.link_generate(n, cp, bps, seed, 3, n_sym, sigma, chan=chan, sym_per_set=per)
    w = wgt if with_eq else float(np.float32(llr_weight(bps, 12.0)))
    ref = oracle.demod(rx, n, cp, bps, eq=eq, weight=(wgt if with_eq else np.float32(w)), sym_per_set=per)
    d_rx, d_bits = torch.from_numpy(rx).cuda(), torch.from_numpy(bits).cuda()
    d_eq = torch.from_numpy(eq).cuda() if with_eq else None
    d_w = torch.from_numpy(wgt).cuda() if with_eq else w

    def run(noalloc):
        llr = torch.full((n_sym, n, bps), 7.0, dtype=torch.float32, device="cuda")
        hard = torch.zeros(n_sym * n * bps // 8, dtype=torch.uint8, device="cuda")
        fft = torch.zeros((n_sym, n), dtype=torch.complex64, device="cuda")
        cnt = torch.zeros(8, dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()
        free0 = torch.cuda.mem_get_info()[0]
        ctx.demod(d_rx, n, cp, bps, eq=d_eq, llr_weight=d_w, sym_per_set=per, out_llr=llr, out_hard=hard, out_fft=fft,
                  ref_hard=d_bits, counters=cnt, noalloc=noalloc)
        # LLR-only form too (another instantiation, no counters: the slots stay untouched)
        llr2 = torch.full((n_sym, n, bps), 9.0, dtype=torch.float32, device="cuda")
        free1 = torch.cuda.mem_get_info()[0]
        ctx.demod(d_rx, n, cp, bps, eq=d_eq, llr_weight=d_w, sym_per_set=per, out_llr=llr2, noalloc=noalloc)
        torch.cuda.synchronize()
        free2 = torch.cuda.mem_get_info()[0]
        return llr.cpu().numpy(), hard.cpu().numpy(), fft.cpu().numpy(), cnt.cpu().numpy(), llr2.cpu().numpy(), (free0, free1, free2)

    a = run(False)
    b = run(True)
    for x, y in zip(a[:5], b[:5]):
        assert np.array_equal(x, y)
    assert np.array_equal(b[0], ref["llr"]) and np.array_equal(b[1], ref["hard"]) and np.array_equal(b[2], ref["fft"])
    assert b[3][0] == oracle.count_bit_errors(ref["hard"], bits)
    assert b[3][1] == n_sym * n * bps and b[3][4] == n_sym
    return b[5]


def test_noalloc_equals_allocating_entry_point_twice(ctx, oracle):
    from sdr_amd.capi import SdrError, SDR_ERR_UNSUPPORTED
    f_first = _case(ctx, oracle, 1024, 256, 4, 37, 11, with_eq=False)
    f_second = _case(ctx, oracle, 256, 64, 6, 9, 12, with_eq=True)     # smaller input, other table, taps
    # a call that has run once allocates nothing: device free memory does not move across either form
    # (torch's llr2 buffer between the two readings is the only allocation, and it is torch's)
    assert f_second[1] - f_second[2] == 0, f_second
    assert f_first[1] - f_first[2] == 0, f_first
    # sizes and layouts beyond the static tables are refused, loudly
    rx = torch.zeros(2 * (16384 + 4096), dtype=torch.complex64, device="cuda")
    with pytest.raises(SdrError) as e:
        ctx.demod(rx, 16384, 4096, 4, noalloc=True)
    assert e.value.code == SDR_ERR_UNSUPPORTED and "8192" in e.value.message
    from sdr_amd.link import noise_sigma
    h_rx, _ = oracle.link_generate(64, 5, 2, 19, 0, 3, np.float32(noise_sigma(64, 2, 6.0)))
    rx = torch.from_numpy(h_rx).cuda()
    with pytest.raises(SdrError) as e:
        ctx.demod(rx, 64, 5, 2, noalloc=True)
    assert e.value.code == SDR_ERR_UNSUPPORTED
    # ... and a refused call leaves the allocating entry point usable: the odd prefix takes the generic
    # engines there, same answers as the oracle
    ref = oracle.demod(h_rx, 64, 5, 2, weight=np.float32(1.5))
    out = ctx.demod(rx, 64, 5, 2, llr_weight=1.5, want_hard=True, want_fft=True)
    torch.cuda.synchronize()
    assert np.array_equal(out.llr.cpu().numpy(), ref["llr"]) and np.array_equal(out.hard.cpu().numpy(), ref["hard"])
    assert np.array_equal(out.fft.cpu().numpy(), ref["fft"])


def test_noalloc_contexts_own_their_counter_slots(ctx, oracle):
    """Each live context owns one of the four static slot sets of its GPU: two contexts on two streams count at
    the same time without disturbing each other, and a context that finds no set free is refused for counting
    calls only (include/sdr_amd.h)."""
    import threading
    from sdr_amd.capi import SdrError, SDR_ERR_UNSUPPORTED
    from sdr_amd.link import noise_sigma
    from sdr_amd.ops import Context
    n, cp, bps, n_sym = 1024, 256, 4, 96
    rx, bits = oracle.link_generate(n, cp, bps, 23, 0, n_sym, np.float32(noise_sigma(n, bps, 8.0)))
    ref = oracle.demod(rx, n, cp, bps, weight=np.float32(1.0), want_fft=False)
    want = oracle.count_bit_errors(ref["hard"], bits)
    results, failures = {}, []

    def worker(tag):
        try:
            stream = torch.cuda.Stream()
            with torch.cuda.stream(stream):
                c = Context(0)
                d_rx, d_bits = torch.from_numpy(rx).cuda(), torch.from_numpy(bits).cuda()
                llr = torch.empty((n_sym, n, bps), dtype=torch.float32, device="cuda")
                hard = torch.empty(n_sym * n * bps // 8, dtype=torch.uint8, device="cuda")
                cnt = torch.zeros(8, dtype=torch.int64, device="cuda")
                for _ in range(25):
                    c.demod(d_rx, n, cp, bps, llr_weight=1.0, out_llr=llr, out_hard=hard, ref_hard=d_bits, counters=cnt, noalloc=True)
                stream.synchronize()
                results[tag] = (hard.cpu().numpy(), cnt.cpu().numpy())
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
        hard, cnt = results[tag]
        assert np.array_equal(hard, ref["hard"])
        assert cnt[0] == 25 * want and cnt[4] == 25 * n_sym and cnt[1] == 25 * n_sym * n * bps
    # `ctx` (the fixture) holds one set; three more contexts take the rest, the next one has none
    import gc
    gc.collect()                                  # contexts other tests dropped without close() give their sets back
    extra = [Context(0) for _ in range(4)]
    try:
        d_rx, d_bits = torch.from_numpy(rx).cuda(), torch.from_numpy(bits).cuda()
        cnt = torch.zeros(8, dtype=torch.int64, device="cuda")
        llr = torch.empty((n_sym, n, bps), dtype=torch.float32, device="cuda")
        with pytest.raises(SdrError) as e:
            extra[3].demod(d_rx, n, cp, bps, out_llr=llr, ref_hard=d_bits, counters=cnt, noalloc=True)
        assert e.value.code == SDR_ERR_UNSUPPORTED and "slot" in e.value.message
        extra[3].demod(d_rx, n, cp, bps, llr_weight=1.0, out_llr=llr, noalloc=True)       # not counting: fine
        extra[0].demod(d_rx, n, cp, bps, llr_weight=1.0, out_llr=llr, ref_hard=d_bits, counters=cnt, noalloc=True)
        torch.cuda.synchronize()
        assert np.array_equal(llr.cpu().numpy(), ref["llr"]) and int(cnt[0].item()) == want
    finally:
        for c in extra:
            c.close()
    # a destroyed context gives its set back
    again = Context(0)
    cnt = torch.zeros(8, dtype=torch.int64, device="cuda")
    again.demod(d_rx, n, cp, bps, llr_weight=1.0, out_llr=llr, ref_hard=d_bits, counters=cnt, noalloc=True)
    torch.cuda.synchronize()
    assert int(cnt[0].item()) == want
    again.close()


def test_noalloc_under_graph_capture(ctx, oracle):
    """The reason the entry point exists: it can be captured into a HIP graph (the allocating one may
    call hipMalloc on a first use, which capture forbids).  Replay equals the eager result."""
    from sdr_amd.link import noise_sigma
    n, cp, bps, n_sym = 512, 128, 4, 16
    rx, bits = oracle.link_generate(n, cp, bps, 77, 0, n_sym, np.float32(noise_sigma(n, bps, 10.0)))
    d_rx, d_bits = torch.from_numpy(rx).cuda(), torch.from_numpy(bits).cuda()
    llr = torch.zeros((n_sym, n, bps), dtype=torch.float32, device="cuda")
    hard = torch.zeros(n_sym * n * bps // 8, dtype=torch.uint8, device="cuda")
    cnt = torch.zeros(8, dtype=torch.int64, device="cuda")
    from sdr_amd.ops import Context
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        c2 = Context(0)                       # enqueues on s
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            c2.demod(d_rx, n, cp, bps, llr_weight=1.25, out_llr=llr, out_hard=hard, ref_hard=d_bits, counters=cnt, noalloc=True)
        for _ in range(3):
            g.replay()
    torch.cuda.synchronize()
    ref = oracle.demod(rx, n, cp, bps, weight=np.float32(1.25))
    assert np.array_equal(llr.cpu().numpy(), ref["llr"]) and np.array_equal(hard.cpu().numpy(), ref["hard"])
    assert int(cnt[0].item()) == 3 * oracle.count_bit_errors(ref["hard"], bits) and int(cnt[4].item()) == 3 * n_sym
    c2.close()
