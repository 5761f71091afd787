"""Process-group plumbing shared by bench.py and ber_sweep.py: one process per GPU, launched by
`python -m torch.distributed.run`, collectives over RCCL (torch's "nccl" backend on ROCm).

The data path has no collective (OFDM symbols are independent, lib/inc/ofdm.hpp:37-43 keeps no
state); what crosses xGMI is one int64 all-reduce(SUM) of the counters per sweep and one float64
all-reduce(MAX) of the elapsed time per measurement.

A process started under torch.distributed.run (RANK / WORLD_SIZE / MASTER_PORT in the environment)
ALWAYS joins a process group and runs the collectives, also when the world is one rank: the RCCL
path is then exercised on a one-GPU box exactly as on a node (tests/test_gpu_sweep.py).  A plain
`python bench.py` has no group and skips them.

SDR_BENCH_BACKEND=gloo is a rehearsal mode for boxes with fewer GPUs than ranks: ranks share the
devices round-robin and the collectives go over gloo on host copies.
"""
from __future__ import annotations

import os


class Group:
    """rank / world / local_rank of this process and the three collectives the drivers use."""

    def __init__(self, torch):
        self.torch = torch
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        launched = all(k in os.environ for k in ("RANK", "WORLD_SIZE", "MASTER_PORT"))
        self.active = launched or self.world > 1
        self.backend = None
        self.dist = None
        self.collectives = 0                 # collectives this process has run (reported in the JSON lines)
        n_dev = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if not self.active:
            self.local_rank = 0
            if n_dev:
                torch.cuda.set_device(0)
            return
        import torch.distributed as dist
        self.dist = dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        self.backend = os.environ.get("SDR_BENCH_BACKEND", "nccl")
        if self.backend == "nccl":
            torch.cuda.set_device(self.local_rank)
            dist.init_process_group("nccl", rank=self.rank, world_size=self.world, device_id=torch.device("cuda", self.local_rank))
        else:
            if n_dev:
                self.local_rank = self.local_rank % n_dev
                torch.cuda.set_device(self.local_rank)
            dist.init_process_group(self.backend, rank=self.rank, world_size=self.world)

    @property
    def on_device(self) -> bool:
        return self.backend == "nccl"

    def barrier(self):
        if self.active:
            self.dist.barrier()
            self.collectives += 1

    def all_reduce_sum_(self, counters):
        """In-place sum of an int64 device tensor over the ranks: the sweep's only collective, O(KiB)."""
        if not self.active:
            return counters
        if self.on_device:
            self.dist.all_reduce(counters, op=self.dist.ReduceOp.SUM)
        else:
            host = counters.cpu()
            self.dist.all_reduce(host, op=self.dist.ReduceOp.SUM)
            counters.copy_(host)
        self.collectives += 1
        return counters

    def max_over_ranks(self, value: float) -> float:
        if not self.active:
            return float(value)
        t = self.torch.tensor([value], dtype=self.torch.float64, device=("cuda" if self.on_device else "cpu"))
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        self.collectives += 1
        return float(t.item())

    def describe(self) -> dict:
        return {"process_group": bool(self.active), "backend": ("rccl (torch 'nccl')" if self.on_device else self.backend), "world": self.world,
                "collectives_run": self.collectives}

    def close(self):
        if self.active:
            self.dist.barrier()
            self.dist.destroy_process_group()
            self.active = False
