"""sdr_placement_search_c32 and sdr_amd.placement.Arena: the offsets it returns lie in the arena, on the 2 MiB grain, apart
from each other; the receive path run on views at those offsets is bit-equal to the oracle (the placement changes where the
bytes lie, never what they are); bad arguments are refused.  No counterpart in the reference (it has no device memory)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

GRAIN = 2 << 20


@pytest.mark.parametrize("n,cp,bps,n_sym,arena_mib,step_mib", [(256, 64, 6, 8192, 512, 0), (1024, 256, 4, 2048, 256, 16), (64, 16, 2, 65536, 384, 32), (2048, 512, 8, 512, 192, 0)])
def test_search_returns_disjoint_grain_aligned_offsets_and_demod_there_equals_oracle(ctx, oracle, n, cp, bps, n_sym, arena_mib, step_mib):
    from sdr_amd.link import noise_sigma
    from sdr_amd.placement import Arena
    arena = Arena(ctx, arena_mib << 20, step_bytes=step_mib << 20)
    rx, llr = arena.place_demod(n, cp, bps, n_sym)
    p = arena.last
    in_bytes, out_bytes = n_sym * (n + cp) * 8, n_sym * n * bps * 4
    assert p["in_offset"] % GRAIN == 0 and p["out_offset"] % GRAIN == 0
    assert p["in_offset"] + in_bytes <= arena.n_bytes and p["out_offset"] + out_bytes <= arena.n_bytes
    assert p["in_offset"] + in_bytes <= p["out_offset"] or p["out_offset"] + out_bytes <= p["in_offset"]
    assert 0.0 < p["model_worst_gbs"] <= p["model_best_gbs"] < 8000.0
    assert rx.data_ptr() == arena.mem.data_ptr() + p["in_offset"] and llr.data_ptr() == arena.mem.data_ptr() + p["out_offset"]
    # the receive path on the placed views: a sample of symbols against the oracle, every bit
    h_rx, _ = oracle.link_generate(n, cp, bps, 77, 0, n_sym, np.float32(noise_sigma(n, bps, 14.0)))
    rx.copy_(torch.from_numpy(h_rx).cuda().view(n_sym, n + cp))
    llr.fill_(float("nan"))
    ctx.demod(rx, n, cp, bps, llr_weight=1.25, out_llr=llr)
    torch.cuda.synchronize()
    pick = np.unique(np.concatenate([np.arange(8), np.linspace(0, n_sym - 1, 24).astype(np.int64)]))
    ref = oracle.demod(h_rx.reshape(n_sym, n + cp)[pick].reshape(-1), n, cp, bps, weight=np.float32(1.25), want_fft=False)
    got = llr.cpu().numpy()
    assert np.array_equal(got[pick].view(np.uint32), ref["llr"].reshape(len(pick), n, bps).view(np.uint32))
    assert not np.isnan(got).any()
    # nothing else may be placed over them; other tensors go into the gaps
    with pytest.raises(ValueError):
        arena.take(p["out_offset"] + GRAIN, (16,), torch.float32)
    with pytest.raises(ValueError):
        arena.place_demod(n, cp, bps, n_sym)
    t = arena.alloc((1024, n), torch.complex64)
    lo, hi = t.data_ptr() - arena.mem.data_ptr(), t.data_ptr() - arena.mem.data_ptr() + t.numel() * 8
    assert lo % GRAIN == 0 and hi <= arena.n_bytes
    for a, b in ((p["in_offset"], p["in_offset"] + in_bytes), (p["out_offset"], p["out_offset"] + out_bytes)):
        assert hi <= a or b <= lo
    with pytest.raises(MemoryError):
        arena.alloc((arena.n_bytes,), torch.uint8)
    arena.reset()
    assert arena.alloc((arena.n_bytes,), torch.uint8).data_ptr() == arena.mem.data_ptr()


def test_search_refuses_what_it_cannot_place(ctx):
    from sdr_amd.capi import SdrError
    arena = torch.empty((64 << 20,), dtype=torch.uint8, device="cuda")
    with pytest.raises(SdrError, match="arena of"):
        ctx.placement_search(arena, 1024, 256, 6, 1 << 16)                  # 0.6 + 1.5 GiB into 64 MiB
    with pytest.raises(SdrError):
        ctx.placement_search(arena, 1000, 256, 6, 64)                       # not a power of two
    with pytest.raises(SdrError):
        ctx.placement_search(arena, 32, 8, 2, 64)                           # below the tuned kernels' range
    with pytest.raises(SdrError):
        ctx.placement_search(arena, 1024, 255, 6, 64)                       # odd prefix
    with pytest.raises(SdrError):
        ctx.placement_search(arena, 1024, 256, 3, 64)                       # no such modulation
    with pytest.raises(SdrError):
        ctx.placement_search(arena, 1024, 256, 6, 0)
    with pytest.raises(SdrError, match="aligned"):
        ctx.placement_search(arena[4:], 1024, 256, 6, 64)
    i, o, best, worst = ctx.placement_search(arena, 1024, 256, 6, 64)       # the smallest of jobs still gets an answer
    assert i % GRAIN == 0 and o % GRAIN == 0 and i != o and best >= worst > 0.0
