"""The N>1 path of bench.py (one process per GPU, independent videos, barrier-only) exercised on CPU with
the gloo backend and world_size 2: sharding, barrier + MAX-over-ranks timing, aggregate value."""
import os
import socket
import sys
import time

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import bench
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    shard = bench.shard_for_rank(rank, world)
    work = 0.10 * (rank + 1)                       # rank 1 is the slow one
    dt = bench.timed_region(lambda: time.sleep(work), lambda: None, dist, None)
    # every rank must report the slowest rank's time, and the job-level value is the sum of the work / that time
    units = torch.tensor([1000.0 * (rank + 1)], dtype=torch.float64)
    dist.all_reduce(units)
    q.put((rank, shard, dt, float(units.item())))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_barrier_and_max_timing():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, s0, t0, u0), (r1, s1, t1, u1) = res
    assert s0 == [0] and s1 == [1]                 # one independent video per rank, no overlap
    assert abs(t0 - t1) < 1e-9                     # both ranks agree on the MAX
    assert 0.19 < t0 < 1.5                         # ... which is the slow rank's time
    assert u0 == u1 == 3000.0


def test_sharding_covers_all_videos_once():
    sys.path.insert(0, ROOT)
    import bench
    for world in (1, 2, 4, 8):
        seen = sorted(v for r in range(world) for v in bench.shard_for_rank(r, world, 8))
        assert seen == list(range(8))
