"""Multi-GPU layout of the hot path: independent sequence shards, one process per GPU.

Tracked sequences never interact (the reference already shards them over worker processes with
no communication: ``lib/test/evaluation/running.py:105-112,183-186``, ``gpu_id = worker_id %
num_gpu``), so the data path has NO collective.  The only exchange is the per-step result record
``(B_local, 5) = [x, y, w, h, confidence]`` gathered to every rank (RCCL ``all_gather`` over xGMI
on the GPU box, ``gloo`` in the CPU tests) -- 5 KB per GPU at B=256, latency-only, issued
asynchronously so it overlaps the next step.
"""
from __future__ import annotations

from typing import List, Sequence


def shard_sequences(n_sequences: int, rank: int, world: int) -> List[int]:
    """Sequence s runs on rank ``s % world`` (the reference's ``worker_id % num_gpu`` rule)."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world {world}")
    return list(range(rank, n_sequences, world))


def shard_sizes(n_sequences: int, world: int) -> List[int]:
    return [len(range(r, n_sequences, world)) for r in range(world)]


class ResultGather:
    """Double-buffered asynchronous all_gather of per-step result records.

    ``submit(step, local)`` copies ``local`` (B_local, 5) into slot ``step & 1`` and starts the
    collective; ``collect(step)`` waits for it and returns the records in GLOBAL sequence order
    (undoing the ``s % world`` interleave).  Ragged shards are padded to the largest shard.
    """

    def __init__(self, n_sequences: int, group=None, device="cpu"):
        import torch
        import torch.distributed as dist
        self.dist, self.torch = dist, torch
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.n = n_sequences
        self.sizes = shard_sizes(n_sequences, self.world)
        self.bmax = max(self.sizes)
        self.local = [torch.zeros(self.bmax, 5, device=device) for _ in range(2)]
        self.all = [torch.zeros(self.world * self.bmax, 5, device=device) for _ in range(2)]
        self.work = [None, None]

    def submit(self, step: int, local):
        k = step & 1
        if self.work[k] is not None:
            self.work[k].wait()
        b = self.sizes[self.rank]
        if local.shape != (b, 5):
            raise ValueError(f"rank {self.rank} expects ({b}, 5) records, got {tuple(local.shape)}")
        self.local[k][:b].copy_(local)
        self.work[k] = self.dist.all_gather_into_tensor(self.all[k], self.local[k], group=self.group, async_op=True)

    def collect(self, step: int):
        k = step & 1
        if self.work[k] is not None:
            self.work[k].wait()
            self.work[k] = None
        g = self.all[k].view(self.world, self.bmax, 5)
        out = self.torch.empty(self.n, 5, device=g.device)
        for r in range(self.world):
            out[r::self.world] = g[r, :self.sizes[r]]
        return out


def merge_rank_results(per_rank: Sequence, n_sequences: int):
    """Pure-python reference of the interleave ``collect`` undoes (used by the tests)."""
    world = len(per_rank)
    out = [None] * n_sequences
    for r in range(world):
        for j, s in enumerate(range(r, n_sequences, world)):
            out[s] = per_rank[r][j]
    return out
