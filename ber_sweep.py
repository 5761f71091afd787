#!/usr/bin/env python3
"""ber_sweep.py -- Eb/N0 Monte-Carlo BER sweep on the device (BASELINE.json configs[4]).

Frames are generated, sent through AWGN, demodulated and counted entirely on the GPU, every
(modulation, Eb/N0) point of the sweep in ONE launch (sdr_link_sweep_c32: nothing but eight counters
per point touches HBM).  The Gaussian samples come from a 24-bit uniform (sdr_amd/csrc/rng.hpp), so
|z| <= 5.77 sigma: points with a BER below about 1e-8 would read low; the default sweep stays above.  With N ranks the frame
range of every (modulation, Eb/N0) point is split over the ranks and ONE all-reduce(sum) of the
int64 counter vector ends the sweep; random streams are keyed by the global symbol index, so
the counts are identical for any number of GPUs.

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


# VALU-issue cost of ONE OFDM symbol in the fused Monte-Carlo kernel at n_fft = 1024, in SIMD cycles:
# instruction-class counts from rocprofv3 PMC passes (profiles/r2_pmc_instruction_classes_mc_final_65536sym.txt) weighted with the issue cost tools/valubench measured on
# MI355X (profiles/r2_valubench_issue_cost.txt: VOP1/VOP2 2.15, v_mad_u64_u32 4.6, transcendental 10 cycles
# per wave64 instruction; the library is built without packed fp32, which would cost 4.1).  Key: bits per subcarrier.
VALU_ISSUE_CYCLES_PER_SYMBOL_1024 = {1: 9487, 2: 10519, 4: 11933, 6: 13765, 8: 16914, 10: 22251}
VALU_LANE_INSTR_PER_SYMBOL_1024 = {1: 261448, 2: 292172, 4: 333080, 6: 386437, 8: 479018, 10: 636723}
# the same for the hard-only bodies (SDR_POINT_HARD_ONLY; profiles/r2_pmc_instruction_classes_mc_hard_only_65536sym.txt)
VALU_ISSUE_CYCLES_PER_SYMBOL_1024_HARD = {1: 9182, 2: 9765, 4: 9901, 6: 10045, 8: 10193, 10: 10391}
VALU_LANE_INSTR_PER_SYMBOL_1024_HARD = {1: 253544, 2: 270908, 4: 274954, 6: 279218, 8: 283632, 10: 289512}
SIMD_CYCLES_PER_S = 256 * 4 * 2.4e9      # 256 CUs x 4 SIMDs at the 2.4 GHz peak engine clock (MI355X_MICROARCH.md)


def valu_roofline(points, counters, n_fft, elapsed, world, hard_only=False):
    """The sweep kernel touches no HBM beyond its counters: its roofline is the VALU issue rate."""
    if n_fft != 1024 or hard_only and not VALU_ISSUE_CYCLES_PER_SYMBOL_1024_HARD:
        return None
    cycles = lanes = 0.0
    cyc_tab = VALU_ISSUE_CYCLES_PER_SYMBOL_1024_HARD if hard_only else VALU_ISSUE_CYCLES_PER_SYMBOL_1024
    lane_tab = VALU_LANE_INSTR_PER_SYMBOL_1024_HARD if hard_only else VALU_LANE_INSTR_PER_SYMBOL_1024
    for p, r in zip(points, counters):
        cycles += r[4] * cyc_tab[p.mod_bits]      # counter 4 = OFDM symbols of the point
        lanes += r[4] * lane_tab[p.mod_bits]
    achieved = cycles / elapsed
    return {"bound": "valu", "achieved": achieved / 1e12, "peak": world * SIMD_CYCLES_PER_S / 1e12, "unit": "T SIMD issue cycles/s",
            "frac": achieved / (world * SIMD_CYCLES_PER_S), "lane_instructions_per_s": lanes / elapsed,
            "note": "achieved = sum over points of symbols x measured VALU issue cycles per symbol; the shader clock under this load is 1.9-2.4 GHz, so frac understates the busy share by up to 20 %",
            "traffic": None}


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
    args = ap.parse_args()

    import torch
    from sdr_amd import link
    from sdr_amd.ops import Context

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # SDR_BENCH_BACKEND=gloo: rehearsal on a box with fewer GPUs than ranks (ranks share devices,
        # the counter all-reduce goes over gloo on host copies); real runs use RCCL.
        backend = os.environ.get("SDR_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            local_rank = local_rank % torch.cuda.device_count()
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend, rank=rank, world_size=world)
    else:
        torch.cuda.set_device(0)
        local_rank = 0
    ctx = Context(local_rank)

    ebn0 = {}
    for m in args.mods:
        top = DEFAULT_MAX_DB.get(m, 12)
        n = int(round(top / args.step_db)) + 1
        ebn0[m] = [i * args.step_db for i in range(n)]
    points = link.plan_sweep(args.mods, ebn0, args.n_fft, args.bits)
    counters = torch.zeros((len(points), 8), dtype=torch.int64, device="cuda")

    # this rank's share of every point: (modulation, sigma, LLR weight, first global symbol, symbols)
    mine = link.rank_points(points, args.n_fft, rank, world)

    def run_once():
        counters.zero_()
        if args.per_point:
            for i, (mod, sigma, w, g0, n_sym) in enumerate(mine):
                if n_sym:
                    ctx.link_ber(args.n_fft, args.cp, mod, args.seed, g0, n_sym, sigma, w, counters=counters[i])
        else:
            ctx.link_sweep(args.n_fft, args.cp, args.seed, mine, counters, hard_only=args.hard_only)      # ONE launch for the whole sweep
        if world > 1:
            if dist.get_backend() == "nccl":
                dist.all_reduce(counters, op=dist.ReduceOp.SUM)  # the only collective: O(KiB) of int64 over RCCL
            else:
                host = counters.cpu()
                dist.all_reduce(host, op=dist.ReduceOp.SUM)
                counters.copy_(host)

    run_once()                                                   # warm-up (twiddle tables, kernels)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.repeat):
        run_once()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = (time.perf_counter() - t0) / args.repeat
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=("cuda" if dist.get_backend() == "nccl" else "cpu"))
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    c = counters.cpu().tolist()
    if rank == 0:
        total_bits = sum(r[1] for r in c)
        total_frames = sum(p.n_frames for p in points)
        print(f"{'mod':>4} {'Eb/N0':>6} {'bits':>12} {'errors':>10} {'BER':>11} {'theory':>11} {'LLR sign err':>12}", file=sys.stderr)
        for p, r in zip(points, c):
            ber = r[0] / r[1] if r[1] else float("nan")
            print(f"{p.mod_bits:>4} {p.ebn0_db:>6.1f} {r[1]:>12d} {r[0]:>10d} {ber:>11.3e} {link.theory_ber(p.mod_bits, p.ebn0_db):>11.3e} {r[6]:>12d}", file=sys.stderr)
        print(json.dumps({
            "metric": "Monte-Carlo BER sweep throughput (generate + AWGN + demodulate + count on device)",
            "value": total_frames / elapsed, "unit": "frames/s", "bits_per_s": total_bits / elapsed,
            "n_gpus": world, "seconds_per_sweep": elapsed, "points": len(points), "total_bits": total_bits,
            "config": {"workload": "ebn0_sweep", "n_fft": args.n_fft, "cp": args.cp, "mods": args.mods, "frames": total_frames, "symbols_per_frame": link.SYMBOLS_PER_FRAME},
            "scaling": "strong", "collective": "one all_reduce(SUM) of %d int64" % (len(points) * 8),
            "launches_per_sweep": (len(points) if args.per_point else 1),
            "checksum_bit_errors": sum(r[0] for r in c),
            "dtype": "f32", "hard_only": bool(args.hard_only),
            "roofline": valu_roofline(points, c, args.n_fft, elapsed, world, args.hard_only),
            "counters": c,
        }))
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
