"""The faded link in resident chunks (sdr_link_chain_c32, BASELINE.json configs[2]): channel
realisations, zero-forcing taps, generator, equalised demodulation and counters equal the oracle's
chain; results do not depend on the chunk size; the generator stream and the demodulation stream
hand the two sample buffers over correctly when there are many chunks."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_channel_realisations_bit_exact(ctx, oracle):
    for n, first, sets in ((64, 0, 3), (1024, 7, 5), (2048, (1 << 33) + 5, 2)):
        got = ctx.link_channel(n, 3, first, sets).cpu().numpy()
        assert np.array_equal(got, oracle.link_channel(n, 3, first, sets))
    h = ctx.link_channel(4096, 9, 0, 64).cpu().numpy()
    assert abs(np.mean(np.abs(h) ** 2) - 1.0) < 0.02 and abs(h.mean()) < 0.01          # unit variance, zero mean


@pytest.fixture(params=["one_kernel", "two_kernels"])
def chain_form(request, ctx):
    """The chunked link runs as ONE kernel per chunk (chain_fused_kernel: samples in LDS only) or as generator + receive kernel
    with the samples in the workspace; the library picks per shape, the knob forces either.  Same results, bit for bit."""
    ctx.lib.sdr_diag_set_tuning(b"chain_form", 2 if request.param == "two_kernels" else 1)
    yield request.param
    ctx.lib.sdr_diag_set_tuning(b"chain_form", 0)


@pytest.mark.parametrize("n,cp,bps,per,n_sym", [(256, 64, 6, 14, 70), (1024, 256, 4, 14, 42), (2048, 512, 6, 14, 28), (64, 16, 2, 3, 50), (4096, 1024, 8, 14, 14),
                                                (128, 32, 10, 14, 30), (512, 128, 1, 7, 23), (8192, 2048, 6, 2, 5)])
def test_chain_equals_oracle_for_any_chunking(ctx, oracle, chain_form, n, cp, bps, per, n_sym):
    from sdr_amd.link import noise_sigma
    ebn0 = {1: 5.0, 2: 8.0, 4: 12.0, 6: 18.0, 8: 24.0, 10: 30.0}[bps]
    sigma = np.float32(noise_sigma(n, bps, ebn0))
    inv_n0 = np.float32(bps * 10 ** (ebn0 / 10))
    g0 = 5 * per
    want, want_llr = oracle.link_chain(n, cp, bps, 3, g0, n_sym, per, sigma, inv_n0, want_llr=True)
    assert want[0] > 0 and want[4] == n_sym
    for chunk in (per, 2 * per, ((n_sym + per - 1) // per) * per, 1000 * per):
        cnt, llr, ws = ctx.link_chain(n, cp, bps, 3, g0, n_sym, chunk, per, float(sigma), float(inv_n0))
        torch.cuda.synchronize()
        assert np.array_equal(cnt.cpu().numpy().astype(np.uint64), want), chunk
        eff = min(chunk, ((n_sym + per - 1) // per) * per)
        last = n_sym - ((n_sym - 1) // eff) * eff
        assert llr.shape[0] == last and np.array_equal(llr.cpu().numpy(), want_llr[n_sym - last:])
    cnt2, llr2, _ = ctx.link_chain(n, cp, bps, 3, g0, n_sym, per, per, float(sigma), float(inv_n0), want_llr=False)
    assert llr2 is None and np.array_equal(cnt2.cpu().numpy().astype(np.uint64), want)


def test_chain_many_chunks_equal_one_chunk_and_reuse_the_workspace(ctx, chain_form):
    """FFT-2048 / 64-QAM / 14-symbol frames at 18 dB like configs[2], 16 chunks against one."""
    from sdr_amd.link import noise_sigma
    n, cp, bps, per = 2048, 512, 6, 14
    n_sym, chunk = 16 * 1792, 1792
    sigma, inv_n0 = noise_sigma(n, bps, 18.0), bps * 10 ** 1.8
    one, _, _ = ctx.link_chain(n, cp, bps, 3, 0, n_sym, n_sym, per, sigma, inv_n0, want_llr=False)
    need = ctx.link_chain_workspace(n, cp, bps, n_sym, chunk, per)
    ws = torch.empty(need, dtype=torch.uint8, device="cuda")
    many = torch.zeros(8, dtype=torch.int64, device="cuda")
    for _ in range(2):                                                # the second pass reuses buffers the first left dirty
        many.zero_()
        ctx.link_chain(n, cp, bps, 3, 0, n_sym, chunk, per, sigma, inv_n0, workspace=ws, counters=many)
    torch.cuda.synchronize()
    assert np.array_equal(one.cpu().numpy(), many.cpu().numpy())
    c = one.cpu().numpy()
    assert c[4] == n_sym and c[1] == n_sym * n * bps and 0 < c[0] < 0.2 * c[1]
    assert need < ctx.link_chain_workspace(n, cp, bps, n_sym, n_sym, per)


def test_chain_argument_errors(ctx):
    from sdr_amd.capi import SdrError, SDR_ERR_INVALID, SDR_ERR_UNSUPPORTED
    with pytest.raises(SdrError) as e:
        ctx.link_chain(1024, 256, 4, 1, 0, 28, 20, 14, 0.1, 10.0)              # chunk not a multiple of the set
    assert e.value.code == SDR_ERR_INVALID
    with pytest.raises(SdrError) as e:
        ctx.link_chain(1024, 256, 4, 1, 3, 28, 14, 14, 0.1, 10.0)              # first symbol inside a set
    assert e.value.code == SDR_ERR_INVALID
    with pytest.raises(SdrError) as e:
        ctx.link_chain(16384, 0, 4, 1, 0, 28, 14, 14, 0.1, 10.0)
    assert e.value.code == SDR_ERR_UNSUPPORTED
    with pytest.raises(SdrError) as e:
        ctx.link_chain(1024, 256, 4, 1, 0, 28, 14, 14, 0.1, 10.0, workspace=torch.empty(4096, dtype=torch.uint8, device="cuda"))
    assert e.value.code == SDR_ERR_INVALID and "workspace" in e.value.message


def test_chain_forms_agree_at_size(ctx):
    """configs[2]'s shape (FFT-2048, 64-QAM, 14-symbol frames, 18 dB) and configs[3]'s (FFT-4096, 256-QAM) over 57 344 / 14 336 symbols:
    one kernel per chunk and generator + receive kernel count the same errors and leave the same LLRs in the workspace."""
    from sdr_amd.link import noise_sigma
    for n, cp, bps, ebn0, n_sym, chunk in ((2048, 512, 6, 18.0, 57344, 14336), (4096, 1024, 8, 24.0, 14336, 7168)):
        sigma, inv_n0 = noise_sigma(n, bps, ebn0), bps * 10 ** (ebn0 / 10)
        res = []
        for form in (1, 2):
            ctx.lib.sdr_diag_set_tuning(b"chain_form", form)
            cnt, llr, _ = ctx.link_chain(n, cp, bps, 11, 28, n_sym, chunk, 14, sigma, inv_n0)
            torch.cuda.synchronize()
            res.append((cnt.cpu().numpy().copy(), llr.clone()))
        ctx.lib.sdr_diag_set_tuning(b"chain_form", 0)
        assert np.array_equal(res[0][0], res[1][0]) and torch.equal(res[0][1].view(torch.int32), res[1][1].view(torch.int32))
        c = res[0][0]
        assert c[4] == n_sym and c[1] == n_sym * n * bps and 0 < c[0] < 0.1 * c[1]
