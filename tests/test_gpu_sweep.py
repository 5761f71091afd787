"""The whole Eb/N0 sweep as one launch (sdr_link_sweep_c32) and the sweep under several ranks:
counters bit-equal to one Monte-Carlo call per point, to the oracle, and independent of how the
frames are split over processes."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("n,cp", [(64, 16), (256, 64), (1024, 256), (2048, 512), (32, 8)])
def test_sweep_one_launch_equals_per_point_calls(ctx, oracle, n, cp):
    from sdr_amd.link import noise_sigma, llr_weight
    rng = np.random.default_rng(n)
    mods = [2, 4, 6, 8, 1, 10, 4, 2]
    points, g0 = [], 7
    for i, m in enumerate(mods):
        db = {1: 3.0, 2: 4.0, 4: 8.0, 6: 13.0, 8: 18.0, 10: 23.0}[m] + float(rng.integers(0, 3))
        n_sym = 0 if i == 3 else int(rng.integers(1, 40 if n >= 1024 else 200))      # one empty point, ragged sizes
        points.append((m, noise_sigma(n, m, db), llr_weight(m, db), g0, n_sym))
        g0 += n_sym + int(rng.integers(0, 5))
    got = ctx.link_sweep(n, cp, 31, points)
    again = ctx.link_sweep(n, cp, 31, points, got.clone())                             # accumulates
    torch.cuda.synchronize()
    got = got.cpu().numpy()
    assert np.array_equal(again.cpu().numpy(), 2 * got)
    for i, (m, sigma, w, first, n_sym) in enumerate(points):
        if n_sym == 0:
            assert not got[i].any()
            continue
        one = ctx.link_ber(n, cp, m, 31, first, n_sym, sigma, w).cpu().numpy()
        assert np.array_equal(got[i], one), (i, m)
        want = oracle.link_ber(n, cp, m, 31, first, n_sym, np.float32(sigma), np.float32(w))
        assert np.array_equal(got[i].astype(np.uint64), want), (i, m)
        assert got[i][1] == n_sym * n * m and got[i][4] == n_sym and got[i][0] > 0


@pytest.mark.parametrize("n,cp", [(64, 16), (256, 64), (1024, 256), (2048, 512), (32, 8)])
def test_sweep_hard_only_points(ctx, oracle, n, cp):
    """SDR_POINT_HARD_ONLY (the reference's own chain: ofdm::rx -> from_constl, no LLRs): counters 0-4
    equal the full form's and the oracle's, counters 5 and 6 are left exactly as they were; flagged and
    unflagged points mix in one launch.  n = 32 takes the generic engine."""
    from sdr_amd.link import noise_sigma, llr_weight
    from sdr_amd.capi import POINT_HARD_ONLY
    rng = np.random.default_rng(100 + n)
    mods = [1, 2, 4, 6, 8, 10, 6, 4]
    points, g0 = [], 3
    for i, m in enumerate(mods):
        db = {1: 3.0, 2: 4.0, 4: 8.0, 6: 13.0, 8: 18.0, 10: 23.0}[m] + float(rng.integers(0, 3))
        n_sym = int(rng.integers(1, 30 if n >= 1024 else 150))
        points.append((m, noise_sigma(n, m, db), llr_weight(m, db), g0, n_sym))
        g0 += n_sym
    full = ctx.link_sweep(n, cp, 77, points).cpu().numpy()
    start = torch.zeros((len(points), 8), dtype=torch.int64, device="cuda")
    start[:, 5], start[:, 6] = 1234, 99                      # must survive a hard-only point
    hard = ctx.link_sweep(n, cp, 77, points, start.clone(), hard_only=True).cpu().numpy()
    mixed_pts = [pt + ((POINT_HARD_ONLY if i % 2 else 0),) for i, pt in enumerate(points)]
    mixed = ctx.link_sweep(n, cp, 77, mixed_pts).cpu().numpy()
    for i, (m, sigma, w, first, n_sym) in enumerate(points):
        want = oracle.link_ber(n, cp, m, 77, first, n_sym, np.float32(sigma), np.float32(w))
        assert np.array_equal(full[i].astype(np.uint64), want), (i, m)
        assert np.array_equal(hard[i][:5], full[i][:5]), (i, m)
        assert hard[i][5] == 1234 and hard[i][6] == 99 and hard[i][7] == 0
        if i % 2:
            assert np.array_equal(mixed[i][:5], full[i][:5]) and mixed[i][5] == 0 and mixed[i][6] == 0
        else:
            assert np.array_equal(mixed[i], full[i])
    with pytest.raises(RuntimeError):
        ctx.link_sweep(n, cp, 77, [points[0] + (2,)])       # unknown flag bit


def test_sweep_argument_errors(ctx):
    from sdr_amd.capi import SdrError, SDR_ERR_INVALID, SDR_ERR_NOT_POW2
    with pytest.raises(SdrError) as e:
        ctx.link_sweep(1024, 256, 1, [(3, 0.1, 1.0, 0, 14)])
    assert e.value.code == SDR_ERR_INVALID
    with pytest.raises(SdrError) as e:
        ctx.link_sweep(1000, 256, 1, [(4, 0.1, 1.0, 0, 14)])
    assert e.value.code == SDR_ERR_NOT_POW2
    assert not ctx.link_sweep(1024, 256, 1, []).numel()
    assert not ctx.link_sweep(1024, 256, 1, [(4, 0.1, 1.0, 0, 0)]).any()


def test_large_sweep_is_split_invariant(ctx):
    """1e8 bits over 12 points: one launch for everything equals the sum of three launches that
    split every point's symbols 3 ways (what three ranks would run)."""
    from sdr_amd import link
    pts = link.plan_sweep([2, 4, 6, 8], {2: [2, 6], 4: [4, 8, 10], 6: [8, 12, 14], 8: [14, 18, 20, 22]}, 1024, 1e8)
    whole = ctx.link_sweep(1024, 256, 5, link.rank_points(pts, 1024, 0, 1)).cpu().numpy()
    parts = sum(ctx.link_sweep(1024, 256, 5, link.rank_points(pts, 1024, r, 3)).cpu().numpy() for r in range(3))
    assert np.array_equal(whole, parts)
    for p, row in zip(pts, whole):
        assert row[1] == p.n_frames * link.SYMBOLS_PER_FRAME * 1024 * p.mod_bits
        ber, th = row[0] / row[1], link.theory_ber(p.mod_bits, p.ebn0_db)
        assert 0.6 * th < ber < 1.6 * th, (p, ber, th)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _run_sweep(ranks, extra):
    env = dict(os.environ)
    env["SDR_BENCH_BACKEND"] = "gloo"            # ranks share device 0; the counter all-reduce goes over gloo
    base = [os.path.join(ROOT, "ber_sweep.py"), "--bits", "6e7", "--n-fft", "512", "--cp", "128", "--mods", "2", "4", "6", "--step-db", "5"] + extra
    if ranks == 1:
        cmd = [sys.executable] + base
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={ranks}", "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port())] + base
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("{")][-1]
    return json.loads(line)


def test_sweep_under_several_ranks_on_one_gpu():
    """The GPU path under N ranks without N GPUs: 3 fresh processes share device 0, each runs its
    frame share of every point in one launch, one all-reduce (gloo here, RCCL on a real node) sums
    the counters.  Identical to the single-process sweep, and to the per-point form."""
    one = _run_sweep(1, [])
    three = _run_sweep(3, [])
    assert three["n_gpus"] == 3 and three["launches_per_sweep"] == 1
    assert three["counters"] == one["counters"]
    assert one["checksum_bit_errors"] > 0 and one["total_bits"] >= 6e7
    per_point = _run_sweep(1, ["--per-point"])
    assert per_point["counters"] == one["counters"]


def _torchrun_one_rank(script, args, timeout=420):
    """A fresh child under torch.distributed.run with ONE rank and the default backend: the process joins a
    process group over RCCL (torch's "nccl") and runs every collective of the multi-GPU path on this GPU."""
    env = {k: v for k, v in os.environ.items() if k != "SDR_BENCH_BACKEND"}
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, script)] + args
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    return json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])


def test_sweep_rccl_path_with_one_rank():
    """The RCCL code path itself -- init_process_group("nccl", device_id=...), the int64 all_reduce(SUM) of the
    counters, the float64 all_reduce(MAX) of the elapsed time, the barriers -- executed on the one GPU of this
    box: counters equal those of the plain single-process sweep."""
    args = ["--bits", "6e7", "--n-fft", "512", "--cp", "128", "--mods", "2", "4", "6", "--step-db", "5"]
    plain = _run_sweep(1, [])
    rccl = _torchrun_one_rank("ber_sweep.py", args)
    pg = rccl["process_group"]
    assert pg["process_group"] and pg["backend"].startswith("rccl") and pg["world"] == 1
    assert pg["collectives_run"] >= 5               # two sweeps' all-reduces, barriers, the elapsed-time maximum
    assert rccl["counters"] == plain["counters"] and rccl["checksum_bit_errors"] == plain["checksum_bit_errors"] > 0
    assert plain["process_group"]["process_group"] is False
    assert rccl["fixed_overhead_us"] > 0 and plain["fixed_overhead_us"] > 0
    assert rccl["gaussian_tail"]["max_sigma"] == 6.66


def test_bench_rccl_path_with_one_rank():
    """bench.py the way the driver starts it for N > 1 (torch.distributed.run, RCCL), with one rank: barrier and
    max-over-ranks run through the process group, and the line keeps the contract's keys."""
    line = _torchrun_one_rank("bench.py", ["--gpus", "1", "--steps", "3", "--warmup", "1", "--symbols", "8192", "--no-cpu", "--no-extra"])
    pg = line["extra"]["process_group"]
    assert pg["process_group"] and pg["backend"].startswith("rccl") and pg["collectives_run"] >= 3
    assert line["n_gpus"] == 1 and line["steps"] == 3 and line["value"] > 0 and line["unit"] == "OFDM symbols/s"
    assert line["roofline"]["bound"] == "hbm" and 0 < line["roofline"]["frac"] < 1
    assert line["per_launch"]["launches"] == 3 and line["per_launch"]["ms_min"] <= line["per_launch"]["ms_median"] <= line["per_launch"]["ms_max"]


def test_sweep_zero_flag_and_cached_point_table(ctx):
    """SDR_SWEEP_ZERO_COUNTERS clears the rows instead of accumulating; a repeated sweep reuses the device copy of its
    point table and a different one replaces it."""
    from sdr_amd.link import noise_sigma, llr_weight
    pts = [(m, noise_sigma(256, m, db), llr_weight(m, db), g0, n) for m, db, g0, n in ((2, 4.0, 0, 30), (6, 12.0, 100, 17), (4, 8.0, 200, 9))]
    once = ctx.link_sweep(256, 64, 3, pts).cpu().numpy()
    dirty = torch.full((3, 8), 12345, dtype=torch.int64, device="cuda")
    assert np.array_equal(ctx.link_sweep(256, 64, 3, pts, dirty, zero_counters=True).cpu().numpy(), once)
    assert np.array_equal(ctx.link_sweep(256, 64, 3, pts, dirty).cpu().numpy(), 2 * once)              # same table again: accumulates
    other = [pts[1], pts[0]]
    swapped = ctx.link_sweep(256, 64, 3, other).cpu().numpy()
    assert np.array_equal(swapped, once[[1, 0]])
    assert np.array_equal(ctx.link_sweep(256, 64, 3, pts, dirty, zero_counters=True).cpu().numpy(), once)
    with pytest.raises(RuntimeError):
        from sdr_amd import capi
        import ctypes
        a = capi.SweepArgs()
        a.struct_size = ctypes.sizeof(capi.SweepArgs)
        a.n_fft, a.cp, a.n_points, a.seed, a.flags = 256, 64, 0, 1, 4
        capi.check(ctx.lib.sdr_link_sweep_c32(ctx._h, ctypes.byref(a)))
