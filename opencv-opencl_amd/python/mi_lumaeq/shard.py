"""Frame -> GPU sharding: frames are independent, so frame k goes to rank k mod N
and no data-path collective exists.  Reference analogue: the unordered worker pool of
OpenCVequalHist.cpp:397-402 (N threads popping one queue); unlike the reference, results are
re-sequenced by frame index.

Used by bench.py (one process per GPU under torch.distributed) and by the gloo CPU tests.
"""
from __future__ import annotations

from typing import Iterable, List, Sequence


def frames_for_rank(n_frames: int, rank: int, world: int) -> List[int]:
    """Indices of the frames rank `rank` processes: k with k mod world == rank (round-robin)."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    return list(range(rank, n_frames, world))


def owner_of(frame_index: int, world: int) -> int:
    return frame_index % world


def merge_in_order(per_rank: Sequence[Sequence], n_frames: int) -> list:
    """Inverse of frames_for_rank: interleave per-rank result lists back into frame order."""
    world = len(per_rank)
    out = [None] * n_frames
    for r, items in enumerate(per_rank):
        idx = frames_for_rank(n_frames, r, world)
        if len(items) != len(idx):
            raise ValueError(f"rank {r} returned {len(items)} results for {len(idx)} frames")
        for k, v in zip(idx, items):
            out[k] = v
    return out


def reduce_over_ranks(value: float, dist=None, op: str = "max") -> float:
    """One scalar reduced over the ranks ("max" or "sum") through the job's own backend; the value itself without a process group.
    A one-rank process group is NOT short-cut: `bench.py --force-dist` brings one up on RCCL precisely so that these statements run
    on one GPU the way they run on eight."""
    if dist is None or not dist.is_available() or not dist.is_initialized():
        return float(value)
    import torch
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([value], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX if op == "max" else dist.ReduceOp.SUM)
    return float(t.item())


def max_over_ranks(seconds: float, dist=None) -> float:
    """Whole-job time of a sharded step = the slowest rank's time (no collective on the data path;
    this single scalar all-reduce only aggregates the measurement)."""
    if dist is None or not dist.is_available() or not dist.is_initialized():
        return float(seconds)
    import torch
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([seconds], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
