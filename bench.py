#!/usr/bin/env python3
"""bench.py -- throughput of the fused receive path (CP removal -> FFT -> [equalise ->] max-log LLR).

A "step" is one pass of the hot path over one batch of synthetic OFDM symbols that is already
resident in HBM (generated on the device by the library's own frame generator + AWGN).
Default workload = BASELINE.json configs[1]: FFT-1024, cp 256, 16-QAM soft LLR, 1 Mi symbols per
GPU.  With N GPUs every rank processes its own batch (weak scaling; OFDM symbols are independent,
so there is no data-path collective); rank 0 prints ONE JSON line.

  python bench.py --gpus 1 --steps 10 --warmup 3
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling

WORKLOADS = {
    # name: (n_fft, cp, bits/subcarrier, symbols per GPU, Eb/N0 dB, per-subcarrier equaliser)
    "fft1024_16qam_1M": (1024, 256, 4, 1 << 20, 12.0, False),      # configs[1] -- the bench line
    "fft1024_64qam_1M": (1024, 256, 6, 1 << 20, 16.0, False),      # north_star target shape without taps
    "fft1024_64qam_eq": (1024, 256, 6, 1 << 20, 16.0, True),       # north_star target: FFT -> equalise -> LLR, taps + weights per 14-symbol frame
    "fft2048_64qam_eq": (2048, 512, 6, 1 << 18, 18.0, True),       # configs[2] shape, one resident chunk
    "fft4096_256qam": (4096, 1024, 8, 1 << 17, 24.0, False),       # configs[3] shape, per-GPU chunk
    "fft64_qpsk": (64, 16, 2, 1 << 22, 10.0, False),               # configs[0] shape on the GPU
}


TRAFFIC_FILES = ("r3_traffic.json", "r2_traffic.json", "r1_traffic.json")      # newest first


def kernel_source_fingerprint():
    """sha256 over the kernel sources (sdr_amd/csrc/*.hpp, *.hip): written into profiles/rN_traffic.json by the PMC
    passes and compared here, so that the bench line says whether its `traffic` figure was measured on THIS build."""
    import glob
    import hashlib
    h = hashlib.sha256()
    for p in sorted(glob.glob(os.path.join(ROOT, "sdr_amd", "csrc", "*.h*"))):
        h.update(os.path.basename(p).encode())
        h.update(open(p, "rb").read())
    return h.hexdigest()[:16]


def measured_traffic(workload, n_sym):
    """(HBM bytes per launch, file, same build?) from the committed PMC passes (profiles/r3_traffic.json, else an
    earlier round's: FETCH_SIZE and WRITE_SIZE collected in separate rocprofv3 runs of this script by
    tools/profile_bench.sh, gfx950 correction applied); (None, None, None) when no pass covers the workload."""
    for name in TRAFFIC_FILES:
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                d = json.load(f)
            k = d["kernels"].get(workload)
            if k is not None:
                return k["hbm_bytes_per_symbol"] * n_sym, name, d.get("kernel_source_fingerprint") == kernel_source_fingerprint()
        except (OSError, ValueError, KeyError):
            continue
    return None, None, None


def algorithmic_bytes_per_symbol(n_fft, bps):
    """payload samples read (CP skipped) + fp32 LLR written."""
    return 8 * n_fft + 4 * n_fft * bps


def make_inputs(ctx, torch, n_fft, cp, bps, n_sym, ebn0_db, with_eq, seed, first_symbol, rx=None):
    from sdr_amd.link import noise_sigma, llr_weight, SYMBOLS_PER_FRAME
    sigma = noise_sigma(n_fft, bps, ebn0_db)
    w0 = llr_weight(bps, ebn0_db)
    if rx is None:
        rx = torch.empty((n_sym, n_fft + cp), dtype=torch.complex64, device="cuda")
    chan = eq = wgt = None
    per = 0
    if with_eq:
        per = SYMBOLS_PER_FRAME
        n_sets = (n_sym + per - 1) // per + 1
        g = torch.Generator(device="cuda").manual_seed(seed)
        chan = torch.complex(torch.randn((n_sets, n_fft), device="cuda", generator=g), torch.randn((n_sets, n_fft), device="cuda", generator=g)) * (0.5 ** 0.5)
        chan = chan.to(torch.complex64).contiguous()
        n0 = 1.0 / (bps * 10 ** (ebn0_db / 10))
        eq, wgt = ctx.zf_taps(chan.view(-1), bps, 1.0 / n0)
    chunk = 1 << 15
    if with_eq:
        chunk = (chunk // per) * per
    for s0 in range(0, n_sym, chunk):
        n = min(chunk, n_sym - s0)
        ch = None
        if with_eq:
            ch = chan[s0 // per:]
        ctx.link_generate(n_fft, cp, bps, seed, first_symbol + s0, n, sigma, chan=ch, sym_per_set=per, want_bits=False, out_rx=rx[s0:s0 + n])
    torch.cuda.synchronize()
    return rx, eq, wgt, w0, per


def cpu_baseline(n_fft, cp, bps, budget_s=12.0, threads=1):
    """The CPU oracle (port of the reference path: ofdm::rx + nearest, plus the same LLR
    definition) on a bounded sample of the same workload.  threads=1 is how the reference runs
    (its OpenMP pragma is dead); threads>1 splits the symbols over host threads."""
    import numpy as np
    from oracle import binding as oracle
    from sdr_amd.link import noise_sigma, llr_weight
    n_s = max(64, min(2048, (1 << 21) // n_fft)) * (1 if threads == 1 else 4)
    rx, _ = oracle.link_generate(n_fft, cp, bps, 2, 0, n_s, np.float32(noise_sigma(n_fft, bps, 12.0)))
    w = np.float32(llr_weight(bps, 12.0))
    oracle.demod(rx[:8], n_fft, cp, bps, weight=w, want_fft=False)
    done, t0 = 0, time.perf_counter()
    while True:
        oracle.demod(rx, n_fft, cp, bps, weight=w, want_fft=False, threads=threads)
        done += n_s
        dt = time.perf_counter() - t0
        if dt >= budget_s:
            break
    return {"value": done / dt, "unit": "OFDM symbols/s", "cores": threads, "kind": "port",
            "sample": f"{done} symbols ({n_s}-symbol batch x {done // n_s} passes) of FFT-{n_fft}/{bps}-bit LLR+hard demod in {dt:.1f} s, {threads} thread(s) of {os.cpu_count()} host CPUs"}


def launch_stats(launch_ms, bytes_per_launch):
    """Per-launch spread of one timed region (HIP events between consecutive launches): placement and box effects
    move a kernel by a few points, so the average is reported with its minimum, median and maximum."""
    v = sorted(launch_ms)
    med = v[len(v) // 2] if len(v) % 2 else 0.5 * (v[len(v) // 2 - 1] + v[len(v) // 2])
    frac = lambda ms: bytes_per_launch / (ms / 1e3) / 1e9 / HBM_PEAK_GBS
    return {"launches": len(v), "ms_min": v[0], "ms_median": med, "ms_max": v[-1], "ms_avg": sum(v) / len(v),
            "hbm_frac_best": frac(v[0]), "hbm_frac_median": frac(med), "hbm_frac_worst": frac(v[-1])}


def make_arena(ctx, torch, want_gib):
    """One block of device memory for the sample and LLR buffers of the receive workloads, which sdr_placement_search_c32
    then places inside it (where the two buffers lie decides 0.74 / 0.77 / 0.80+ of the HBM peak for the
    same kernel).  None when the device cannot spare 64 GiB: the workloads then use plain allocations."""
    from sdr_amd.placement import Arena
    free, _ = torch.cuda.mem_get_info()
    n_bytes = min(int(want_gib * (1 << 30)), free - (24 << 30))      # room for taps, weights, tables and the allocator's own blocks
    if n_bytes < (64 << 30):
        return None
    try:
        return Arena(ctx, n_bytes)
    except torch.cuda.OutOfMemoryError:
        return None


def time_workload(ctx, torch, group, name, steps, warmup, n_sym_override=None, arena=None):
    n_fft, cp, bps, n_sym, ebn0, with_eq = WORKLOADS[name]
    if n_sym_override:
        n_sym = n_sym_override
    world, rank = group.world, group.rank
    first = rank * ((n_sym + 13) // 14) * 14      # whole frames per rank, disjoint global symbol ranges
    rx = llr = placement = None
    if arena is not None and 8 * (n_fft + cp) * n_sym + 4 * n_fft * bps * n_sym + (8 << 20) <= arena.n_bytes:
        arena.reset()
        try:
            rx, llr = arena.place_demod(n_fft, cp, bps, n_sym)      # the search writes into the arena: inputs are generated after it
        except Exception as e:                                      # a benchmark line with plain allocations beats none
            print(f"placement search failed, plain allocations instead: {e}", file=sys.stderr)
            rx = llr = None
    if rx is not None:
        placement = {"arena_GiB": arena.n_bytes / (1 << 30), "rx_offset_GiB": arena.last["in_offset"] / (1 << 30), "llr_offset_GiB": arena.last["out_offset"] / (1 << 30),
                     "traffic_model_GBs_at_best_placement": arena.last["model_best_gbs"], "traffic_model_GBs_at_worst_placement": arena.last["model_worst_gbs"],
                     "what": "sdr_placement_search_c32: sample and LLR buffers as views of one arena, at the offsets where a traffic model of the kernel ran fastest"}
        llr.zero_()
    rx, eq, wgt, w0, per = make_inputs(ctx, torch, n_fft, cp, bps, n_sym, ebn0, with_eq, seed=2, first_symbol=first, rx=rx)
    if llr is None:
        llr = torch.zeros((n_sym, n_fft, bps), dtype=torch.float32, device="cuda")

    def step():
        ctx.demod(rx, n_fft, cp, bps, eq=eq, llr_weight=(wgt if wgt is not None else w0), sym_per_set=per, out_llr=llr)

    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    group.barrier()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    t0 = time.perf_counter()
    ev[0].record()
    for i in range(steps):
        step()
        ev[i + 1].record()
    torch.cuda.synchronize()
    group.barrier()
    torch.cuda.synchronize()
    elapsed = group.max_over_ranks(time.perf_counter() - t0)
    # the memory system's rate for this traffic shape: same bytes, same access pattern, no arithmetic
    copy_gbs = None
    if eq is None:
        for _ in range(2):
            ctx.copy_shape(rx, n_fft, cp, bps, llr)
        c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        c0.record()
        for _ in range(max(3, steps // 2)):
            ctx.copy_shape(rx, n_fft, cp, bps, llr)
        c1.record()
        torch.cuda.synchronize()
        copy_gbs = algorithmic_bytes_per_symbol(n_fft, bps) * n_sym / (c0.elapsed_time(c1) / max(3, steps // 2) / 1e3) / 1e9
    launch_ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(steps)]
    avg_launch_s = sum(launch_ms) / len(launch_ms) / 1e3
    b_sym = algorithmic_bytes_per_symbol(n_fft, bps)
    achieved = b_sym * n_sym / avg_launch_s / 1e9
    # Taps (8 B) and weights (4 B) per bin are read from HBM once per set of `per` symbols.  SURVEY.md
    # 8(d) leaves them out of the algorithmic count ("amortised / L2-resident"), so `frac` does too;
    # the bytes are real (2.7 % at 14 symbols per set) and are reported beside it.
    tap_bytes_sym = (12.0 * n_fft / per) if with_eq else 0.0
    traffic, traffic_file, traffic_same_build = measured_traffic(name, n_sym)
    res = {
        "n_fft": n_fft, "cp": cp, "bps": bps, "n_sym": n_sym, "elapsed": elapsed, "with_eq": with_eq,
        "tap_bytes_per_symbol": tap_bytes_sym, "frac_incl_taps": (b_sym + tap_bytes_sym) * n_sym / avg_launch_s / 1e9 / HBM_PEAK_GBS,
        "sym_per_s": world * n_sym * steps / elapsed,
        "copy_shape_gbs": copy_gbs,
        "per_launch": launch_stats(launch_ms, b_sym * n_sym),
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "traffic": traffic, "traffic_unit": f"bytes per launch (rocprofv3 PMC, separate passes; profiles/{traffic_file})",
                     "traffic_measured_on_this_build": traffic_same_build,
                     "algorithmic_bytes_per_launch": b_sym * n_sym, "bytes_per_symbol": b_sym, "symbols_per_launch": n_sym, "avg_launch_ms": avg_launch_s * 1e3},
    }
    res["placement"] = placement or {"what": "plain allocations (torch's caching allocator)"}
    res["buffers"] = {k: hex(v.data_ptr()) for k, v in (("rx", rx), ("llr", llr), ("eq", eq), ("wgt", wgt)) if v is not None}      # placement, for the record
    del rx, llr, eq, wgt
    torch.cuda.empty_cache()
    return res


def time_demod_at_stated_size(ctx, torch, name, passes=2, arena=None):
    """BASELINE.json configs[2] at its STATED size, demodulator alone: all 4 Mi symbols (86 GB of samples) and their
    taps and weights (7.4 GB) are generated once, untimed, and stay resident; the timed region is the equalised LLR
    kernel over every chunk of symbols, each launch writing the one 12.9 GB LLR buffer (206 GB of LLRs do not fit
    beside the samples).  Nothing but the demodulator runs inside the timed region."""
    from sdr_amd.link import noise_sigma
    c = dict(CHAIN[name])
    n_fft, cp, bps, per, n_sym, chunk = c["n_fft"], c["cp"], c["bps"], c["per"], c["n_sym"], c["chunk"]
    sigma = noise_sigma(n_fft, bps, c["ebn0"])
    inv_n0 = bps * 10 ** (c["ebn0"] / 10)
    n_sets = (n_sym + per - 1) // per
    llr = placement = None
    resident = 8 * (n_fft + cp) * n_sym + 12 * n_fft * n_sets + 4 * n_fft * bps * chunk
    if arena is not None and resident + (64 << 20) <= arena.n_bytes:
        # the LLR buffer where a chunk's output runs fastest; the 86 GB of samples cannot all lie at one best offset and take the
        # largest gap beside it, taps and weights the next
        arena.reset()
        p = arena.search(n_fft, cp, bps, chunk)
        placement = {"arena_GiB": arena.n_bytes / (1 << 30), "traffic_model_GBs_at_best_placement": p["model_best_gbs"], "traffic_model_GBs_at_worst_placement": p["model_worst_gbs"]}
        try:
            llr = arena.take(p["out_offset"], (chunk, n_fft, bps), torch.float32)
            rx = arena.alloc((n_sym, n_fft + cp), torch.complex64)
            taps = arena.alloc((n_sets, n_fft), torch.complex64)
            wgt = arena.alloc((n_sets, n_fft), torch.float32)
        except MemoryError:                                # the gaps beside the best LLR offset do not hold the samples: one after the other
            arena.reset()
            rx = arena.alloc((n_sym, n_fft + cp), torch.complex64)
            taps = arena.alloc((n_sets, n_fft), torch.complex64)
            wgt = arena.alloc((n_sets, n_fft), torch.float32)
            llr = arena.alloc((chunk, n_fft, bps), torch.float32)
            placement["what"] = "first fit (the gaps beside the best LLR offset were too small for the samples)"
        placement.update({"llr_offset_GiB": (llr.data_ptr() - arena.mem.data_ptr()) / (1 << 30), "rx_offset_GiB": (rx.data_ptr() - arena.mem.data_ptr()) / (1 << 30)})
        llr.zero_()
    else:
        rx = torch.empty((n_sym, n_fft + cp), dtype=torch.complex64, device="cuda")
        taps = torch.empty((n_sets, n_fft), dtype=torch.complex64, device="cuda")
        wgt = torch.empty((n_sets, n_fft), dtype=torch.float32, device="cuda")
    gen_sets = 4096                                        # 57344 symbols per generator call
    for s0 in range(0, n_sets, gen_sets):
        ns = min(gen_sets, n_sets - s0)
        chan = ctx.link_channel(n_fft, c["seed"], s0, ns)
        t, w = ctx.zf_taps(chan.view(-1), bps, inv_n0)
        taps[s0:s0 + ns] = t.view(ns, n_fft)
        wgt[s0:s0 + ns] = w.view(ns, n_fft)
        first = s0 * per
        cnt = min(ns * per, n_sym - first)
        ctx.link_generate(n_fft, cp, bps, c["seed"], first, cnt, sigma, chan=chan, sym_per_set=per, want_bits=False, out_rx=rx[first:first + cnt])
    del chan, t, w
    torch.cuda.synchronize()
    if llr is None:
        llr = torch.zeros((chunk, n_fft, bps), dtype=torch.float32, device="cuda")
    starts = list(range(0, n_sym, chunk))

    def one_pass(events=None):
        for i, s0 in enumerate(starts):
            n = min(chunk, n_sym - s0)
            ctx.demod(rx[s0:s0 + n], n_fft, cp, bps, eq=taps[s0 // per:], llr_weight=wgt[s0 // per:], sym_per_set=per, out_llr=llr[:n])
            if events is not None:
                events[i + 1].record()

    one_pass()                                             # warm-up: first touch of the LLR buffer, tables
    torch.cuda.synchronize()
    b_sym = algorithmic_bytes_per_symbol(n_fft, bps)
    ms, t0 = [], time.perf_counter()
    for _ in range(passes):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(len(starts) + 1)]
        ev[0].record()
        one_pass(ev)
        torch.cuda.synchronize()
        ms.append([ev[i].elapsed_time(ev[i + 1]) for i in range(len(starts))])
    wall = (time.perf_counter() - t0) / passes
    full = [m for p in ms for i, m in enumerate(p) if min(chunk, n_sym - starts[i]) == chunk]      # full-size chunks
    kernel_s = sum(sum(p) for p in ms) / passes / 1e3
    res = {"symbols": n_sym, "chunk_symbols": chunk, "chunks": len(starts), "resident_GB": (rx.numel() * 8 + taps.numel() * 8 + wgt.numel() * 4 + llr.numel() * 4) / 1e9,
           "seconds_per_pass": kernel_s, "wall_seconds_per_pass": wall, "sym_per_s": n_sym / kernel_s, "llr_gbit_per_s": n_sym / kernel_s * n_fft * bps / 1e9,
           "achieved_GBs": b_sym * n_sym / kernel_s / 1e9, "hbm_frac": b_sym * n_sym / kernel_s / 1e9 / HBM_PEAK_GBS,
           "hbm_frac_incl_tap_bytes": (b_sym + 12.0 * n_fft / per) * n_sym / kernel_s / 1e9 / HBM_PEAK_GBS,
           "per_launch": launch_stats(full, b_sym * chunk),
           "placement": placement or {"what": "plain allocations (torch's caching allocator)"},
           "what": "equalised LLR kernel (taps + weights per 14-symbol frame) over all resident symbols, chunk by chunk into one reused LLR buffer; generation untimed"}
    del rx, taps, wgt, llr
    torch.cuda.empty_cache()
    return res


def time_sweep(ctx, torch, bits=1e10, repeat=3):
    """BASELINE.json configs[4] on this GPU: the default 70-point Eb/N0 grid (QPSK ... 256-QAM, FFT-1024), 1e10 bits,
    ONE sdr_link_sweep_c32 launch per sweep.  ber_sweep.py is the multi-GPU form with the all-reduce."""
    import ber_sweep
    from sdr_amd import link
    mods = [2, 4, 6, 8]
    pts = link.plan_sweep(mods, ber_sweep.default_grid(mods, 1.0), 1024, bits)
    rows = link.rank_points(pts, 1024, 0, 1)
    counters = torch.zeros((len(pts), 8), dtype=torch.int64, device="cuda")
    ctx.link_sweep(1024, 256, 5, rows, counters, zero_counters=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(repeat):
        ctx.link_sweep(1024, 256, 5, rows, counters, zero_counters=True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / repeat
    c = counters.cpu().tolist()
    roofline, util = ber_sweep.valu_figures(pts, c, 1024, dt, 1)
    total_bits = sum(r[1] for r in c)
    worst = max(abs(r[0] / r[1] / link.theory_ber(p.mod_bits, p.ebn0_db) - 1.0) for p, r in zip(pts, c) if r[0] >= 1000)
    return {"points": len(pts), "total_bits": total_bits, "seconds_per_sweep": dt, "bits_per_s": total_bits / dt,
            "frames_per_s": sum(p.n_frames for p in pts) / dt, "symbols_per_s": sum(r[4] for r in c) / dt, "launches_per_sweep": 1,
            "checksum_bit_errors": sum(r[0] for r in c), "largest_deviation_from_closed_form_ber": worst,
            "roofline": roofline, "valu_issue_utilisation": util,
            "what": "generate -> IFFT -> AWGN -> FFT -> demap -> count on the device, nothing but 70 x 8 counters in HBM"}


CHAIN = {
    # BASELINE.json configs[2] at its stated size: FFT-2048, cp 512, 64-QAM, one channel realisation per
    # 14-symbol frame, 18 dB, 4 Mi symbols (86 GB of samples, 206 GB of LLRs) in resident chunks
    "fft2048_64qam_eq_4M_chain": dict(n_fft=2048, cp=512, bps=6, n_sym=1 << 22, chunk=(1 << 18) // 14 * 14, per=14, ebn0=18.0, seed=3),
}


def time_chain(ctx, torch, name, steps, warmup, n_sym_override=None):
    """One step = the whole faded link over n_sym symbols: channel -> zero-forcing taps -> generator +
    AWGN -> equalised demodulation to LLRs -> counters, chunk by chunk in one reused workspace, the
    generator of chunk i+1 on a second stream beside the demodulation of chunk i (sdr_link_chain_c32)."""
    from sdr_amd.link import noise_sigma
    c = dict(CHAIN[name])
    if n_sym_override:
        c["n_sym"] = n_sym_override
    sigma = noise_sigma(c["n_fft"], c["bps"], c["ebn0"])
    inv_n0 = c["bps"] * 10 ** (c["ebn0"] / 10)
    need = ctx.link_chain_workspace(c["n_fft"], c["cp"], c["bps"], c["n_sym"], c["chunk"], c["per"])
    ws = torch.empty(need, dtype=torch.uint8, device="cuda")
    cnt = torch.zeros(8, dtype=torch.int64, device="cuda")

    def step():
        ctx.link_chain(c["n_fft"], c["cp"], c["bps"], c["seed"], 0, c["n_sym"], c["chunk"], c["per"], sigma, inv_n0, workspace=ws, counters=cnt)

    for _ in range(warmup):
        step()
    cnt.zero_()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    k = cnt.cpu().tolist()
    b_sym = algorithmic_bytes_per_symbol(c["n_fft"], c["bps"])
    return {"symbols": c["n_sym"], "chunk_symbols": c["chunk"], "chunks": -(-c["n_sym"] // c["chunk"]), "workspace_GB": need / 1e9,
            "seconds_per_pass": dt, "sym_per_s": c["n_sym"] / dt, "llr_gbit_per_s": c["n_sym"] / dt * c["n_fft"] * c["bps"] / 1e9,
            "demod_bytes_GBs": b_sym * c["n_sym"] / dt / 1e9, "generated_GBs": 8 * (c["n_fft"] + c["cp"]) * c["n_sym"] / dt / 1e9,
            "bit_errors_per_pass": k[0] // steps, "bits_per_pass": k[1] // steps, "ber": (k[0] / k[1]) if k[1] else None,
            "what": "generate (channel + AWGN) and demodulate (taps + weights, LLRs written, hard bits counted) overlapped on two streams, per pass over all symbols"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)      # 50 x 4.4 ms: a 0.2 s timed region
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="fft1024_16qam_1M", choices=sorted(WORKLOADS))
    ap.add_argument("--chain", default="", choices=[""] + sorted(CHAIN), help="only run this chunked whole-link configuration and print its JSON")
    ap.add_argument("--symbols", type=int, default=0, help="override symbols per GPU (debug)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--no-extra", action="store_true", help="skip the extra measurements (other configurations, config 3 at 4 Mi symbols, the Eb/N0 sweep)")
    ap.add_argument("--demod-4m", action="store_true", help="only run configs[2]'s demodulator at its stated 4 Mi symbols and print its JSON")
    ap.add_argument("--sweep", action="store_true", help="only run the one-GPU Eb/N0 sweep (configs[4]) and print its JSON")
    ap.add_argument("--placement", default="search", choices=["search", "default"],
                    help="search: sample and LLR buffers are views of one arena, placed by sdr_placement_search_c32 (DESIGN.md 7.0); default: plain allocations")
    ap.add_argument("--arena-gib", type=float, default=200.0, help="size of the arena for --placement search (clipped to what the device has free)")
    args = ap.parse_args()

    import torch
    from sdr_amd.dist import Group
    group = Group(torch)      # under torch.distributed.run: RCCL process group, also with one rank
    world, rank, local_rank = group.world, group.rank, group.local_rank
    if args.gpus != world and rank == 0 and world > 1:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)

    from sdr_amd.ops import Context
    ctx = Context(local_rank)
    arena = make_arena(ctx, torch, args.arena_gib) if args.placement == "search" and not (args.sweep or args.chain) else None
    if args.demod_4m or args.sweep:
        r = time_demod_at_stated_size(ctx, torch, "fft2048_64qam_eq_4M_chain", arena=arena) if args.demod_4m else time_sweep(ctx, torch)
        if rank == 0:
            print(json.dumps({"config": {"workload": "fft2048_64qam_eq_4M_demod" if args.demod_4m else "ebn0_sweep_1e10"}, **r}))
        ctx.close()
        group.close()
        return
    if args.chain:
        r = time_chain(ctx, torch, args.chain, max(1, args.steps), max(1, min(args.warmup, 2)), args.symbols or None)
        if rank == 0:
            print(json.dumps({"metric": "whole faded link, OFDM symbols/s (generate + equalised demodulation to LLRs + counting, resident chunks)", "value": r["sym_per_s"],
                              "unit": "OFDM symbols/s", "n_gpus": 1, "config": {"workload": args.chain}, **r}))
        ctx.close()
        group.close()
        return
    main_res = time_workload(ctx, torch, group, args.workload, args.steps, args.warmup, args.symbols or None, arena=arena)
    extra = {"process_group": group.describe(), "placement": main_res["placement"]}
    if world == 1 and not args.no_extra and args.workload == "fft1024_16qam_1M" and not args.symbols:
        # the north_star target shape without and with the equaliser ACTIVE, and the other configurations of
        # BASELINE.json's list, for the record (not bench lines); first launches on fresh buffers run 5-15 % slow,
        # hence 10 untimed launches ahead of each timed region
        for name in ("fft1024_64qam_1M", "fft1024_64qam_eq", "fft2048_64qam_eq", "fft4096_256qam", "fft64_qpsk"):
            r = time_workload(ctx, torch, group, name, max(10, args.steps // 2), 10, arena=arena)
            extra[name] = {"sym_per_s": r["sym_per_s"], "llr_gbit_per_s": r["sym_per_s"] * r["n_fft"] * r["bps"] / 1e9,
                           "hbm_frac": r["roofline"]["frac"], "achieved_GBs": r["roofline"]["achieved"], "symbols": r["n_sym"],
                           "equaliser": r["with_eq"], "hbm_traffic_bytes_per_launch": r["roofline"]["traffic"], "per_launch": r["per_launch"], "placement": r["placement"]}
            if r["with_eq"]:
                extra[name]["hbm_frac_incl_tap_bytes"] = r["frac_incl_taps"]
                extra[name]["tap_bytes_per_symbol"] = r["tap_bytes_per_symbol"]
        try:                           # configs[2] at its stated size: the demodulator alone
            extra["fft2048_64qam_eq_4M_demod"] = time_demod_at_stated_size(ctx, torch, "fft2048_64qam_eq_4M_chain", arena=arena)
        except Exception as e:
            if arena is None:
                raise
            print(f"4 Mi-symbol pass inside the arena failed, plain allocations instead: {e}", file=sys.stderr)
            arena.close()
            arena = None
            torch.cuda.empty_cache()
            extra["fft2048_64qam_eq_4M_demod"] = time_demod_at_stated_size(ctx, torch, "fft2048_64qam_eq_4M_chain")
        if arena is not None:          # the chain and the sweep bring their own workspaces
            arena.close()
            arena = None
            torch.cuda.empty_cache()
        extra["fft2048_64qam_eq_4M_chain"] = time_chain(ctx, torch, "fft2048_64qam_eq_4M_chain", 2, 1)               # ... and generated + demodulated + counted
        extra["ebn0_sweep_1e10"] = time_sweep(ctx, torch)                                                            # configs[4] on one GPU
    if main_res.get("copy_shape_gbs"):
        extra["copy_of_same_shape"] = {"achieved_GBs": main_res["copy_shape_gbs"], "kernel_over_copy": main_res["roofline"]["achieved"] / main_res["copy_shape_gbs"],
                                       "what": "sdr_diag_copy_shape_c32: the kernel's bytes and access pattern with no arithmetic, same buffers"}
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu:
        cpu = cpu_baseline(main_res["n_fft"], main_res["cp"], main_res["bps"])
        # the box's CPU share for one GPU is what OMP_NUM_THREADS says (16), not os.cpu_count()
        share = max(1, min(int(os.environ.get("OMP_NUM_THREADS", "0") or 0) or (os.cpu_count() or 1), os.cpu_count() or 1, 64))
        if share > 1:
            extra["cpu_baseline_all_threads"] = cpu_baseline(main_res["n_fft"], main_res["cp"], main_res["bps"], budget_s=6.0, threads=share)
    if rank == 0:
        n_fft, bps = main_res["n_fft"], main_res["bps"]
        line = {
            "metric": "demodulated OFDM symbols/s (fused CP-removal + FFT + " + ("equalise + " if main_res["with_eq"] else "") + "max-log LLR"
                      + ("" if main_res["with_eq"] else "; no equaliser in this workload, see extra.fft1024_64qam_eq") + ")",
            "value": main_res["sym_per_s"],
            "unit": "OFDM symbols/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": main_res["elapsed"] / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic (device frame generator: Philox bits -> QAM -> IFFT+CP -> AWGN)",
            "config": {"workload": args.workload, "n_fft": n_fft, "cp": main_res["cp"], "bits_per_subcarrier": bps,
                       "symbols_per_gpu": main_res["n_sym"], "sharding": f"independent symbol batches x{world}",
                       "buffer_placement": "sdr_placement_search_c32 in one arena" if "arena_GiB" in main_res["placement"] else "plain allocations"},
            "llr_gbit_per_s": main_res["sym_per_s"] * n_fft * bps / 1e9,
            "roofline": main_res["roofline"],
            "per_launch": main_res["per_launch"],
            "cpu_baseline": cpu,
            "extra": extra,
        }
        print(json.dumps(line))
    ctx.close()
    group.close()


if __name__ == "__main__":
    main()
