#!/usr/bin/env python3
"""ber_sweep.py -- Eb/N0 Monte-Carlo BER sweep on the device (BASELINE.json configs[4]).

Frames are generated, sent through AWGN, demodulated and counted entirely on the GPU, every
(modulation, Eb/N0) point of the sweep in ONE launch (sdr_link_sweep_c32: nothing but eight counters
per point touches HBM).  The Gaussian samples come from a 32-bit uniform (sdr_amd/csrc/rng.hpp), so
|z| <= 6.66 sigma (a true Gaussian holds 2.7e-11 per component beyond that): BER estimates down to the
1e-9 that 1e10 bits resolve are unaffected.  With N ranks the frame range of every (modulation, Eb/N0)
point is split over the ranks and ONE all-reduce(sum) of the int64 counter vector ends the sweep; random
streams are keyed by the global symbol index, so the counts are identical for any number of GPUs.
Started under torch.distributed.run the process group (RCCL) is used even with one rank.

  python ber_sweep.py --bits 1e10
  python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 ber_sweep.py --bits 1e10

Prints the BER table (rank 0, stderr) and one JSON line (stdout).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

DEFAULT_MAX_DB = {2: 10, 4: 14, 6: 18, 8: 24}


# VALU-issue cost of ONE OFDM symbol in the fused Monte-Carlo kernel at n_fft = 1024, in SIMD cycles: instruction counts
# from rocprofv3 PMC passes over the round's kernels (tools/r3_mc_pmc.sh, profiles/r3_pmc_instruction_classes_mc_65536sym.txt)
# priced by tools/mc_issue_tables.py with the issue costs tools/valubench measured on MI355X
# (profiles/r2_valubench_issue_cost.txt: VOP1/VOP2 2.15, v_mad_u64_u32 4.6, transcendental 10 cycles per wave64
# instruction; VOP3-encoded instructions, 4.1-4.6 cycles, cannot be told apart by the counters and are priced as VOP2, so
# the figures are lower bounds; the library is built without packed fp32).  Key: bits per subcarrier.
# (Round 2's kernels: 9487 ... 22251 cycles, 261448 ... 636723 lane instructions.)
VALU_ISSUE_CYCLES_PER_SYMBOL_1024 = {1: 8561, 2: 9473, 4: 10563, 6: 12201, 8: 15173, 10: 20235}
VALU_LANE_INSTR_PER_SYMBOL_1024 = {1: 234962, 2: 262098, 4: 294565, 6: 343299, 8: 431790, 10: 582450}
# the same for the hard-only bodies (SDR_POINT_HARD_ONLY; profiles/r3_pmc_instruction_classes_mc_hard_only_65536sym.txt)
VALU_ISSUE_CYCLES_PER_SYMBOL_1024_HARD = {1: 8336, 2: 8887, 4: 8775, 6: 8899, 8: 9037, 10: 9254}
VALU_LANE_INSTR_PER_SYMBOL_1024_HARD = {1: 228352, 2: 244736, 4: 241399, 6: 245082, 8: 249199, 10: 255656}
SIMD_CYCLES_PER_S = 256 * 4 * 2.4e9      # 256 CUs x 4 SIMDs at the 2.4 GHz peak engine clock (MI355X_MICROARCH.md)
# Peak of the vector ALU for unpacked fp32 / int32 work: 157 TFLOP/s of packed FMA = 64 flop per SIMD and cycle, i.e. 32
# lanes of one operation per SIMD and cycle (a wave64 instruction every two cycles; tools/valubench measures 2.15).
VALU_LANE_OPS_PER_S = SIMD_CYCLES_PER_S * 32


def algorithmic_lane_ops_per_symbol(n_fft, bps, hard_only=False):
    """Lane-operations one OFDM symbol of the Monte-Carlo link NEEDS, counted from the definitions (DESIGN.md
    sections 3 and 4), not from the compiled kernel: what is left when every instruction that is not arithmetic of
    the algorithm (addressing, LDS traffic, selects, moves, loop control) is taken away.  One lane-operation = one
    fp32 / int32 operation on one value; -ffp-contract=off makes a multiply-add two.
      transforms   2 x (N/2) log2 N radix-2 butterflies x 10 operations (four products, six sums: fft.hpp:127-137)
      Philox       N/8 index blocks + N/2 noise blocks, 10 rounds x (2 wide multiplies + 2 three-input xors)
      Gaussian     per complex sample (rng.hpp, table-based since round 3): uniform from 32 bits (3), ln = exponent and
                   mantissa split, table pair, cubic log1p (17), -2 ln, clamp, sqrt (3), sincos = split of the 24-bit
                   angle, table pair, two short series and the addition theorems (18), scaling and the addition to
                   the sample (5)
      mapping      index -> two Gray-decoded levels x norm, and the inverse transform's division by N
      demapping    x inorm, slicer per axis (nearest: fused coordinate, clamp, round, miss, convert) and its two tests,
                   and per axis with L = 2^m levels: L squared distances (2 L),
                   3 L - 2 m - 4 minima when the aligned-block minima are shared between the bits of the axis (82 at
                   L = 32, qam.hpp axis_llr_level), one difference and one weight product per bit (exact max-log)
      statistics   per LLR: scaling, half a packed 16-bit conversion, half a dot product, a sign compare; sign errors as one
                   xor + popcount per point
    """
    import math
    log2n = int(round(math.log2(n_fft)))
    transforms = 2 * (n_fft // 2) * log2n * 10
    philox = (max(n_fft // 8, 1) + n_fft // 2) * 10 * 4
    gaussian = n_fft * (3 + 17 + 3 + 18 + 5)
    mapping = n_fft * (6 + 2)
    mi = (bps + 1) // 2 if bps > 1 else 1
    mq = bps - mi
    nearest = n_fft * (4 + sum(6 for m in (mi, mq) if m))
    llr = 0
    stats = n_fft * 3                                  # xor + popcount + symbol-error compare on the hard decision
    if not hard_only:
        for m in (mi, mq):
            if m:
                levels = 1 << m
                llr += n_fft * (2 * levels + max(3 * levels - 2 * m - 4, 0) + 2 * m)
        stats += n_fft * (bps * 3 + 2)
    return transforms + philox + gaussian + mapping + nearest + llr + stats


def valu_figures(points, counters, n_fft, elapsed, world, hard_only=False):
    """The sweep kernel touches no HBM beyond its counters, so it is priced against the vector ALU.
    Returns (roofline, valu_issue_utilisation):
      roofline: achieved = ALGORITHMIC lane-operations per second (algorithmic_lane_ops_per_symbol x symbols / time)
        against the VALU's peak for unpacked 32-bit work -- how far the kernel is from what the algorithm needs;
      valu_issue_utilisation: the kernel's OWN measured instruction counts (PMC, n_fft = 1024 only) times the measured
        issue cost per instruction class, over the SIMD cycles available -- how busy the issue slots are, which
        says nothing about how many of those instructions the algorithm needs."""
    cyc_tab = VALU_ISSUE_CYCLES_PER_SYMBOL_1024_HARD if hard_only else VALU_ISSUE_CYCLES_PER_SYMBOL_1024
    lane_tab = VALU_LANE_INSTR_PER_SYMBOL_1024_HARD if hard_only else VALU_LANE_INSTR_PER_SYMBOL_1024
    cycles = lanes = needed = 0.0
    for p, r in zip(points, counters):
        n_sym = r[4]                                   # counter 4 = OFDM symbols of the point
        needed += n_sym * algorithmic_lane_ops_per_symbol(n_fft, p.mod_bits, hard_only)
        if n_fft == 1024:
            cycles += n_sym * cyc_tab[p.mod_bits]
            lanes += n_sym * lane_tab[p.mod_bits]
    peak = world * VALU_LANE_OPS_PER_S
    roofline = {"bound": "valu", "achieved": needed / elapsed / 1e12, "peak": peak / 1e12, "unit": "T lane-operations/s (unpacked 32-bit)",
                "frac": needed / elapsed / peak, "traffic": None,
                "note": "achieved = algorithmic lane-operations (transform butterflies, Philox rounds, Gaussian, mapping, demapping, statistics; "
                        "ber_sweep.algorithmic_lane_ops_per_symbol) per second; peak = 1024 SIMDs x 32 lanes x 2.4 GHz"}
    util = None
    if n_fft == 1024 and lanes:
        util = {"frac": cycles / elapsed / (world * SIMD_CYCLES_PER_S), "issue_cycles_per_s": cycles / elapsed, "simd_cycles_per_s": world * SIMD_CYCLES_PER_S,
                "executed_lane_instructions_per_s": lanes / elapsed, "algorithmic_over_executed": needed / lanes,
                "note": "kernel's own PMC instruction counts x measured issue cost per class (profiles/r2_valubench_issue_cost.txt; VOP3 forms priced as VOP2, i.e. a lower bound); the shader clock under "
                        "this load is 1.9-2.4 GHz, so frac understates the busy share by up to 20 %; algorithmic_over_executed = how much of what the kernel "
                        "executes the algorithm needs"}
    return roofline, util


# Box-Muller on a 32-bit uniform u1 in [2^-32, 1] (sdr_amd/csrc/rng.hpp): |z| <= sqrt(-2 ln 2^-32) sigma = 6.66 sigma
GAUSSIAN_TAIL_SIGMA = 6.66
GAUSSIAN_TAIL_PROBABILITY = 2.7e-11      # per real component: what a true Gaussian holds beyond the bound


def default_grid(mods, step_db):
    ebn0 = {}
    for m in mods:
        top = DEFAULT_MAX_DB.get(m, 12)
        n = int(round(top / step_db)) + 1
        ebn0[m] = [i * step_db for i in range(n)]
    return ebn0


def fixed_overhead_us(ctx, torch, link, group, n_fft, cp, seed, points, hard_only, reps=200):
    """What one sweep costs beyond its arithmetic: the same sweep with ONE frame per point (70 x 14 symbols, a few
    microseconds of kernel time spread over the chip), timed from the first host call to the end of the
    synchronisation: clearing the counters, building and handing over the point table, the launch, the all-reduce
    when a process group exists, and the wait."""
    import time
    tiny = [link.SweepPoint(p.mod_bits, p.ebn0_db, group.world, i * group.world) for i, p in enumerate(points)]      # one frame per point and rank
    rows = link.rank_points(tiny, n_fft, group.rank, group.world)
    counters = torch.zeros((len(tiny), 8), dtype=torch.int64, device="cuda")

    def once():
        ctx.link_sweep(n_fft, cp, seed, rows, counters, hard_only=hard_only, zero_counters=True)
        group.all_reduce_sum_(counters)
        torch.cuda.synchronize()

    for _ in range(10):
        once()
    group.barrier()
    t0 = time.perf_counter()
    for _ in range(reps):
        once()
    return (time.perf_counter() - t0) / reps * 1e6


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bits", type=float, default=1e10, help="total bits over the whole sweep")
    ap.add_argument("--n-fft", type=int, default=1024)
    ap.add_argument("--cp", type=int, default=256)
    ap.add_argument("--mods", type=int, nargs="+", default=[2, 4, 6, 8])
    ap.add_argument("--step-db", type=float, default=1.0)
    ap.add_argument("--seed", type=int, default=5)
    ap.add_argument("--repeat", type=int, default=1, help="timed repetitions of the sweep")
    ap.add_argument("--hard-only", action="store_true", help="SDR_POINT_HARD_ONLY on every point: BER from hard decisions alone (the reference's chain), no LLR statistics")
    ap.add_argument("--per-point", action="store_true", help="one sdr_link_ber_c32 launch per point (the round-1 form) instead of one sdr_link_sweep_c32 launch per sweep")
    ap.add_argument("--no-overhead", action="store_true", help="skip the fixed-overhead measurement")
    args = ap.parse_args()

    import torch
    from sdr_amd import link
    from sdr_amd.dist import Group
    from sdr_amd.ops import Context

    group = Group(torch)
    rank, world = group.rank, group.world
    ctx = Context(group.local_rank)

    points = link.plan_sweep(args.mods, default_grid(args.mods, args.step_db), args.n_fft, args.bits)
    counters = torch.zeros((len(points), 8), dtype=torch.int64, device="cuda")

    # this rank's share of every point: (modulation, sigma, LLR weight, first global symbol, symbols)
    mine = link.rank_points(points, args.n_fft, rank, world)

    def run_once():
        if args.per_point:
            counters.zero_()
            for i, (mod, sigma, w, g0, n_sym) in enumerate(mine):
                if n_sym:
                    ctx.link_ber(args.n_fft, args.cp, mod, args.seed, g0, n_sym, sigma, w, counters=counters[i])
        else:
            # ONE launch for the whole sweep; the library clears the counter rows on the same stream ahead of it
            ctx.link_sweep(args.n_fft, args.cp, args.seed, mine, counters, hard_only=args.hard_only, zero_counters=True)
        group.all_reduce_sum_(counters)             # the only collective: O(KiB) of int64 over RCCL

    run_once()                                                   # warm-up (twiddle tables, kernels)
    torch.cuda.synchronize()
    group.barrier()
    t0 = time.perf_counter()
    for _ in range(args.repeat):
        run_once()
    torch.cuda.synchronize()
    group.barrier()
    elapsed = group.max_over_ranks((time.perf_counter() - t0) / args.repeat)
    c = counters.cpu().tolist()
    overhead = None
    if not args.no_overhead and not args.per_point:
        overhead = group.max_over_ranks(fixed_overhead_us(ctx, torch, link, group, args.n_fft, args.cp, args.seed, points, args.hard_only))
    if rank == 0:
        total_bits = sum(r[1] for r in c)
        total_frames = sum(p.n_frames for p in points)
        print(f"{'mod':>4} {'Eb/N0':>6} {'bits':>12} {'errors':>10} {'BER':>11} {'theory':>11} {'LLR sign err':>12}", file=sys.stderr)
        for p, r in zip(points, c):
            ber = r[0] / r[1] if r[1] else float("nan")
            print(f"{p.mod_bits:>4} {p.ebn0_db:>6.1f} {r[1]:>12d} {r[0]:>10d} {ber:>11.3e} {link.theory_ber(p.mod_bits, p.ebn0_db):>11.3e} {r[6]:>12d}", file=sys.stderr)
        roofline, util = valu_figures(points, c, args.n_fft, elapsed, world, args.hard_only)
        print(json.dumps({
            "metric": "Monte-Carlo BER sweep throughput (generate + AWGN + demodulate + count on device)",
            "value": total_frames / elapsed, "unit": "frames/s", "bits_per_s": total_bits / elapsed,
            "n_gpus": world, "seconds_per_sweep": elapsed, "points": len(points), "total_bits": total_bits,
            "config": {"workload": "ebn0_sweep", "n_fft": args.n_fft, "cp": args.cp, "mods": args.mods, "frames": total_frames, "symbols_per_frame": link.SYMBOLS_PER_FRAME},
            "scaling": "strong", "collective": "one all_reduce(SUM) of %d int64" % (len(points) * 8), "process_group": group.describe(),
            "launches_per_sweep": (len(points) if args.per_point else 1),
            "fixed_overhead_us": overhead,
            "fixed_overhead_what": "one sweep of one 14-symbol frame per point and rank, first host call to end of synchronisation: counter clear + point table + launch + all-reduce (when a process group exists) + wait",
            "checksum_bit_errors": sum(r[0] for r in c),
            "dtype": "f32", "hard_only": bool(args.hard_only),
            "gaussian_tail": {"max_sigma": GAUSSIAN_TAIL_SIGMA, "true_gaussian_mass_beyond_per_component": GAUSSIAN_TAIL_PROBABILITY,
                              "note": "Box-Muller on a 32-bit uniform: no sample beyond 6.66 sigma; BER floors it could bias lie below 1e-10"},
            "roofline": roofline, "valu_issue_utilisation": util,
            "counters": c,
        }))
    ctx.close()
    group.close()


if __name__ == "__main__":
    main()
