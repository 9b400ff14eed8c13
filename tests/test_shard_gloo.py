"""N>1 path on CPU: world_size-2 gloo processes shard frames (frame k -> rank k mod N), process their
shard independently (no data-path collective) and the results re-sequence to the serial answer.
The compute stand-in here is the oracle (test-only): what is being tested is the sharding, the
in-order merge and the max-over-ranks timing reduction that bench.py uses."""
import os
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parents[1]



def _free_port():
    """A rendezvous port the kernel hands out (bind to port 0), as bench._free_port does: a port derived from the pid can collide with the
    other gloo tests of this file or with a torchrun default on a busy box."""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_frames, q):
    sys.path.insert(0, str(ROOT))
    sys.path.insert(0, str(ROOT / "opencv-opencl_amd" / "python"))
    import oracle
    from mi_lumaeq import shard, synth
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        w, h = 64, 36
        mine = shard.frames_for_rank(n_frames, rank, world)
        sums = []
        for k in mine:
            out = oracle.nv12_frame(synth.nv12_frame(w, h, "D2", k), w, h, uv_mode=1, op=0)
            sums.append(int(out.astype(np.uint64).sum()) * 1000003 + int(out[::7].astype(np.uint64).sum()))
        gathered = [None] * world
        dist.all_gather_object(gathered, sums)          # test-side collection only; the data path has none
        t = shard.max_over_ranks(0.25 * (rank + 1), dist)
        tot = shard.reduce_over_ranks(10.0 + rank, dist, "sum")          # what the bench sums over the ranks (late frames, frames/s)
        mx = shard.reduce_over_ranks(-1.0 if rank else 3.5, dist, "max")
        assert tot == sum(10.0 + r for r in range(world)) and mx == 3.5
        if rank == 0:
            q.put((shard.merge_in_order(gathered, n_frames), t))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_frames", [7, 8])
def test_two_rank_sharding_matches_serial(n_frames):
    sys.path.insert(0, str(ROOT))
    import oracle
    from mi_lumaeq import shard, synth
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_frames, q)) for r in range(2)]
    for p in procs:
        p.start()
    merged, t = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    w, h = 64, 36
    serial = []
    for k in range(n_frames):
        out = oracle.nv12_frame(synth.nv12_frame(w, h, "D2", k), w, h, uv_mode=1, op=0)
        serial.append(int(out.astype(np.uint64).sum()) * 1000003 + int(out[::7].astype(np.uint64).sum()))
    assert merged == serial
    assert t == pytest.approx(0.5)                       # slowest rank


def test_shard_helpers():
    from mi_lumaeq import shard
    for n in (0, 1, 5, 64, 513):
        for world in (1, 2, 4, 8):
            parts = [shard.frames_for_rank(n, r, world) for r in range(world)]
            assert sorted(sum(parts, [])) == list(range(n))
            assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1
            assert all(shard.owner_of(k, world) == r for r, p in enumerate(parts) for k in p)
            assert shard.merge_in_order([[k * 10 for k in p] for p in parts], n) == [k * 10 for k in range(n)]
    with pytest.raises(ValueError):
        shard.frames_for_rank(4, 2, 2)
    assert shard.max_over_ranks(1.5) == 1.5
    assert shard.reduce_over_ranks(2.5, None, "sum") == 2.5


def _one_rank_worker(port, q):
    sys.path.insert(0, str(ROOT / "opencv-opencl_amd" / "python"))
    from mi_lumaeq import shard
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=0, world_size=1)
    calls = []
    real = dist.all_reduce

    def counting(t, *a, **kw):
        calls.append(t.device.type)
        return real(t, *a, **kw)
    dist.all_reduce = counting
    try:
        got = (shard.max_over_ranks(0.75, dist), shard.reduce_over_ranks(4.0, dist, "sum"), shard.reduce_over_ranks(-2.0, dist, "max"))
    finally:
        dist.all_reduce = real
        dist.destroy_process_group()
    q.put((got, calls))


def test_one_rank_group_still_goes_through_the_backend():
    """`bench.py --force-dist` (world size 1 on RCCL) exists so that the reductions of the N > 1 path execute on one GPU: a one-rank
    group must really call all_reduce, not return early."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_one_rank_worker, args=(_free_port(), q))
    p.start()
    got, calls = q.get(timeout=120)
    p.join(timeout=60)
    assert p.exitcode == 0
    assert got == (0.75, 4.0, -2.0) and calls == ["cpu", "cpu", "cpu"]
