"""CPU tests of the host-side driver logic (sdr_amd/link.py) including a world_size-2 gloo run
of the sharded sweep, with the CPU oracle standing in for the GPU counting kernel."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_frames_tiles_range():
    from sdr_amd.link import shard_frames
    for n in (0, 1, 7, 8, 1000, 1001):
        for world in (1, 2, 3, 8):
            spans = [shard_frames(n, r, world) for r in range(world)]
            pos = 0
            for first, count in spans:
                assert first == pos
                pos += count
            assert pos == n
            assert max(c for _, c in spans) - min(c for _, c in spans) <= 1


def test_noise_scaling():
    from sdr_amd.link import noise_sigma, llr_weight, grid_energy
    # per-bin complex noise variance N * 2*sigma^2 must equal N0 = 1/(bps*EbN0)
    for n, bps, db in ((64, 2, 10.0), (1024, 4, 12.0), (4096, 8, 24.0)):
        s = noise_sigma(n, bps, db)
        n0 = 1.0 / (bps * 10 ** (db / 10))
        assert abs(n * 2 * s * s - n0) < 1e-12
        assert abs(llr_weight(bps, db) - 1.0 / (grid_energy(bps) * n0)) < 1e-9


def test_plan_sweep_uses_disjoint_frames():
    from sdr_amd.link import plan_sweep, SYMBOLS_PER_FRAME
    pts = plan_sweep([2, 4], {2: [0, 1, 2], 4: [0, 4]}, 1024, 1e8)
    pos = 0
    for p in pts:
        assert p.first_frame == pos
        pos += p.n_frames
    assert sum(p.n_frames * SYMBOLS_PER_FRAME * 1024 * p.mod_bits for p in pts) >= 1e8


def _oracle_count(n_fft, cp, seed):
    from oracle import binding as o

    def fn(p, g0, n_sym, sigma, w):
        return [int(v) for v in o.link_ber(n_fft, cp, p.mod_bits, seed, g0, n_sym, np.float32(sigma), np.float32(w))]
    return fn


def test_sweep_counts_do_not_depend_on_world_size():
    from sdr_amd.link import plan_sweep, run_sweep
    n_fft, cp, seed = 64, 16, 5
    pts = plan_sweep([2, 4], {2: [2.0, 4.0], 4: [6.0]}, n_fft, 3e5)
    one = run_sweep(pts, n_fft, cp, seed, 0, 1, _oracle_count(n_fft, cp, seed))
    parts = [run_sweep(pts, n_fft, cp, seed, r, 3, _oracle_count(n_fft, cp, seed)) for r in range(3)]
    for i in range(len(pts)):
        summed = [sum(p.counters[i][j] for p in parts) for j in range(8)]
        assert summed == one.counters[i]
        assert one.counters[i][0] > 0


def _gloo_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from sdr_amd.link import plan_sweep, run_sweep, torch_all_reduce
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n_fft, cp, seed = 64, 16, 9
    pts = plan_sweep([2, 6], {2: [3.0], 6: [10.0, 12.0]}, n_fft, 2e5)
    res = run_sweep(pts, n_fft, cp, seed, rank, world, _oracle_count(n_fft, cp, seed), torch_all_reduce)
    q.put((rank, res.counters))
    dist.barrier()
    dist.destroy_process_group()


def test_sweep_gloo_world2_matches_single_rank():
    import torch.multiprocessing as mp
    from sdr_amd.link import plan_sweep, run_sweep
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_gloo_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    n_fft, cp, seed = 64, 16, 9
    pts = plan_sweep([2, 6], {2: [3.0], 6: [10.0, 12.0]}, n_fft, 2e5)
    ref = run_sweep(pts, n_fft, cp, seed, 0, 1, _oracle_count(n_fft, cp, seed))
    assert got[0] == ref.counters and got[1] == ref.counters


def test_rank_points_rows_line_up_and_tile():
    from sdr_amd.link import plan_sweep, rank_points, SYMBOLS_PER_FRAME, noise_sigma
    pts = plan_sweep([2, 4, 8], {2: [0, 1], 4: [3.0], 8: [20.0, 21.0]}, 256, 4e6)
    for world in (1, 2, 3, 8, 64):
        rows = [rank_points(pts, 256, r, world) for r in range(world)]
        for i, p in enumerate(pts):
            pos = p.first_frame * SYMBOLS_PER_FRAME
            for r in range(world):
                mod, sigma, w, g0, n_sym = rows[r][i]
                assert mod == p.mod_bits and sigma == noise_sigma(256, p.mod_bits, p.ebn0_db)
                assert g0 == pos and n_sym % SYMBOLS_PER_FRAME == 0
                pos += n_sym
            assert pos == (p.first_frame + p.n_frames) * SYMBOLS_PER_FRAME



def _group_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update({"RANK": str(rank), "WORLD_SIZE": str(world), "LOCAL_RANK": str(rank), "MASTER_ADDR": "127.0.0.1",
                       "MASTER_PORT": str(port), "SDR_BENCH_BACKEND": "gloo"})
    import torch
    from sdr_amd.dist import Group
    g = Group(torch)
    t = torch.arange(16, dtype=torch.int64).view(2, 8) * (rank + 1)
    g.all_reduce_sum_(t)
    slowest = g.max_over_ranks(1.5 + rank)
    g.barrier()
    q.put((rank, t.tolist(), slowest, g.describe()))
    g.close()


@pytest.mark.parametrize("world", [1, 2])
def test_process_group_helper_over_gloo(world):
    """sdr_amd.dist.Group (what bench.py and ber_sweep.py use): a process started with RANK / WORLD_SIZE / MASTER_PORT
    joins a group and runs the collectives, ALSO when the world is one rank; sums and maxima are what they should be."""
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    procs = [mpc.Process(target=_group_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    factor = sum(r + 1 for r in range(world))
    for rank, t, slowest, desc in got:
        assert t == [[factor * (8 * i + j) for j in range(8)] for i in range(2)]
        assert slowest == 1.5 + (world - 1)
        assert desc["process_group"] is True and desc["backend"] == "gloo" and desc["world"] == world and desc["collectives_run"] == 3


def test_no_process_group_without_launcher(monkeypatch):
    import torch
    from sdr_amd.dist import Group
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        monkeypatch.delenv(k, raising=False)
    g = Group(torch)
    assert not g.active and g.world == 1 and g.rank == 0
    t = torch.ones(4, dtype=torch.int64)
    assert g.all_reduce_sum_(t) is t and g.max_over_ranks(2.5) == 2.5
    g.barrier()
    assert g.describe()["collectives_run"] == 0
    g.close()


def test_algorithmic_operation_count_model():
    """ber_sweep.algorithmic_lane_ops_per_symbol: the transform term is 2 x (N/2) log2 N x 10, the count grows with the
    constellation, the hard-only form needs less, and it stays below what the kernel was measured to execute."""
    import ber_sweep
    for n in (64, 1024, 4096):
        for bps in (1, 2, 4, 6, 8, 10):
            full, hard = ber_sweep.algorithmic_lane_ops_per_symbol(n, bps), ber_sweep.algorithmic_lane_ops_per_symbol(n, bps, True)
            assert full > hard > 2 * (n // 2) * (n.bit_length() - 1) * 10
    for bps, measured in ber_sweep.VALU_LANE_INSTR_PER_SYMBOL_1024.items():
        assert 0.5 * measured < ber_sweep.algorithmic_lane_ops_per_symbol(1024, bps) < measured


def test_arena_bookkeeping_on_cpu_memory():
    """sdr_amd.placement.Arena: views, first-fit gaps, overlap refusal and reset, on a CPU tensor (the search itself needs the GPU:
    tests/test_gpu_placement.py)."""
    import torch
    from sdr_amd.placement import GRAIN, Arena

    class Stub:
        torch_device = torch.device("cpu")

    a = Arena(Stub(), 64 * GRAIN + 12345)
    assert a.n_bytes == 64 * GRAIN                                   # whole grains only
    x = a.take(4 * GRAIN, (GRAIN // 8 + 1,), torch.complex64)        # one grain and eight bytes: two grains taken
    assert x.data_ptr() == a.mem.data_ptr() + 4 * GRAIN and a.taken == [(4 * GRAIN, 6 * GRAIN)]
    with pytest.raises(ValueError):
        a.take(5 * GRAIN, (16,), torch.float32)                      # inside x
    with pytest.raises(ValueError):
        a.take(63 * GRAIN, (GRAIN,), torch.float32)                  # leaves the arena
    y = a.alloc((3 * GRAIN,), torch.uint8)                           # the gap in front of x holds it
    assert y.data_ptr() == a.mem.data_ptr()
    z = a.alloc((2 * GRAIN // 4,), torch.float32)                    # the rest of that gap does not: behind x
    assert z.data_ptr() == a.mem.data_ptr() + 6 * GRAIN
    w = a.alloc((GRAIN,), torch.uint8)                               # one grain is left in front of x
    assert w.data_ptr() == a.mem.data_ptr() + 3 * GRAIN
    with pytest.raises(MemoryError):
        a.alloc((57 * GRAIN,), torch.uint8)                          # 56 grains are left behind z
    assert a.alloc((56 * GRAIN,), torch.uint8).data_ptr() == a.mem.data_ptr() + 8 * GRAIN
    x.view(torch.float32).fill_(1.0)
    assert float(a.mem[4 * GRAIN:4 * GRAIN + 4].view(torch.float32)[0]) == 1.0      # views alias the arena
    a.reset()
    assert a.alloc((64 * GRAIN,), torch.uint8).data_ptr() == a.mem.data_ptr()
    with pytest.raises(ValueError):
        a.place_demod(1024, 256, 4, 16)                              # live views: the search would overwrite them
    a.close()
    assert a.mem is None
