"""Monte-Carlo BER driver: Eb/N0 -> noise scaling, frame sharding over ranks, counter reduction.

No reference counterpart (the reference's only end-to-end loop is the GUI tick,
app/QFDemoWindow.cpp:201-224, with an identity channel).  OFDM frames are independent
(ofdm::rx keeps no state across symbols, lib/inc/ofdm.hpp:37-43), so ranks take disjoint frame
ranges with no data-path collective; the only exchange is one integer all-reduce of the
counters per sweep.  Random streams are keyed by the GLOBAL symbol index, so the summed
counters are identical for any number of ranks.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Callable, Dict, List, Sequence, Tuple

SYMBOLS_PER_FRAME = 14
N_COUNTERS = 8


def grid_energy(bps: int) -> float:
    if bps == 1:
        return 1.0
    levels = float(1 << (bps // 2))
    return 2.0 * (levels * levels - 1.0) / 3.0


def noise_sigma(n_fft: int, bps: int, ebn0_db: float) -> float:
    """Per-component standard deviation of the time-domain noise.

    Es = 1 per subcarrier, Eb = 1/bps, N0 = Eb / (Eb/N0).  ifft2 divides by N and fft2 does not
    (lib/inc/fft.hpp:200-202), so a complex time sample of variance s2 appears in every bin with
    variance N*s2; hence s2 = N0/N and each real component has variance s2/2.
    """
    ebn0 = 10.0 ** (ebn0_db / 10.0)
    n0 = 1.0 / (bps * ebn0)
    return math.sqrt(n0 / (2.0 * n_fft))


def llr_weight(bps: int, ebn0_db: float) -> float:
    """w = norm^2 / N0 for a flat channel (|H| = 1)."""
    ebn0 = 10.0 ** (ebn0_db / 10.0)
    n0 = 1.0 / (bps * ebn0)
    return (1.0 / grid_energy(bps)) / n0


def shard_frames(n_frames: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous frame range [first, first+count) of `rank`; ranges tile [0, n_frames)."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    base, extra = divmod(n_frames, world)
    first = rank * base + min(rank, extra)
    count = base + (1 if rank < extra else 0)
    return first, count


@dataclass
class SweepPoint:
    mod_bits: int
    ebn0_db: float
    n_frames: int           # total over all ranks
    first_frame: int = 0    # global index of the first frame of this point


@dataclass
class SweepResult:
    points: List[SweepPoint]
    counters: List[List[int]] = field(default_factory=list)   # per point, N_COUNTERS integers

    def ber(self, i: int) -> float:
        c = self.counters[i]
        return c[0] / c[1] if c[1] else float("nan")


def plan_sweep(mods: Sequence[int], ebn0_db: Dict[int, Sequence[float]], n_fft: int, total_bits: float) -> List[SweepPoint]:
    """Splits `total_bits` evenly over all (modulation, Eb/N0) points, in whole 14-symbol frames.
    Every point gets its own range of global frame indices so no random stream is reused."""
    n_points = sum(len(ebn0_db[m]) for m in mods)
    bits_per_point = total_bits / max(n_points, 1)
    pts: List[SweepPoint] = []
    next_frame = 0
    for m in mods:
        frame_bits = SYMBOLS_PER_FRAME * n_fft * m
        frames = max(1, int(math.ceil(bits_per_point / frame_bits)))
        for e in ebn0_db[m]:
            pts.append(SweepPoint(m, float(e), frames, next_frame))
            next_frame += frames
    return pts


def rank_points(points: Sequence[SweepPoint], n_fft: int, rank: int, world: int) -> List[Tuple[int, float, float, int, int]]:
    """This rank's share of every point as the rows of an sdr_link_sweep_c32 point table:
    (modulation bits, noise sigma, LLR weight, first global symbol, symbols).  Ranks take contiguous
    frame ranges (shard_frames); a rank without frames of a point keeps the row with 0 symbols, so
    counter rows line up across ranks for the all-reduce."""
    rows = []
    for p in points:
        first, count = shard_frames(p.n_frames, rank, world)
        rows.append((p.mod_bits, noise_sigma(n_fft, p.mod_bits, p.ebn0_db), llr_weight(p.mod_bits, p.ebn0_db),
                     (p.first_frame + first) * SYMBOLS_PER_FRAME, count * SYMBOLS_PER_FRAME))
    return rows


def run_sweep(points: Sequence[SweepPoint], n_fft: int, cp: int, seed: int, rank: int, world: int,
              count_fn: Callable[[SweepPoint, int, int, float, float], Sequence[int]],
              all_reduce_fn: Callable[[List[int]], List[int]] = None) -> SweepResult:
    """Runs this rank's share of every point and reduces once at the end.

    count_fn(point, first_symbol, n_symbols, sigma, llr_weight) -> N_COUNTERS integers for that
    symbol range (on the GPU: Context.link_ber).  all_reduce_fn sums a flat integer list over
    ranks (torch.distributed all_reduce on an int64 tensor; RCCL on GPUs, gloo in CPU tests).
    """
    flat: List[int] = []
    for p in points:
        first, count = shard_frames(p.n_frames, rank, world)
        sigma = noise_sigma(n_fft, p.mod_bits, p.ebn0_db)
        w = llr_weight(p.mod_bits, p.ebn0_db)
        g0 = (p.first_frame + first) * SYMBOLS_PER_FRAME
        c = list(count_fn(p, g0, count * SYMBOLS_PER_FRAME, sigma, w)) if count else [0] * N_COUNTERS
        if len(c) != N_COUNTERS:
            raise ValueError("count_fn must return %d counters" % N_COUNTERS)
        flat.extend(int(v) for v in c)
    if all_reduce_fn is not None and world > 1:
        flat = list(all_reduce_fn(flat))
    res = SweepResult(list(points))
    for i in range(len(points)):
        res.counters.append(flat[i * N_COUNTERS:(i + 1) * N_COUNTERS])
    return res


def torch_all_reduce(flat: List[int]) -> List[int]:
    """One all-reduce(sum) of int64 counters: RCCL ('nccl' backend) on GPUs, gloo on CPU."""
    import torch
    import torch.distributed as dist

    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    t = torch.tensor(flat, dtype=torch.int64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return [int(v) for v in t.cpu().tolist()]


def theory_ber(bps: int, ebn0_db: float) -> float:
    """Textbook AWGN approximations (exact for BPSK/QPSK), used only as sanity bounds in tests."""
    ebn0 = 10.0 ** (ebn0_db / 10.0)
    q = lambda x: 0.5 * math.erfc(x / math.sqrt(2.0))
    if bps <= 2:
        return q(math.sqrt(2.0 * ebn0))
    m = float(1 << bps)
    return (4.0 / bps) * (1.0 - 1.0 / math.sqrt(m)) * q(math.sqrt(3.0 * bps * ebn0 / (m - 1.0)))
