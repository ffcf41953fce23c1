"""The N>1 path of bench.py (one process per GPU, independent videos, barrier-only) exercised on CPU with
the gloo backend and world_size 2: sharding, barrier + MAX-over-ranks timing, aggregate value."""
import os
import socket
import sys
import time

import pytest
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


def _launcher_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import aiod_amd  # noqa: F401
    from aiod_amd import launch_videos as L
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    seen = []

    def fake_stage1(name):            # stands in for stage1.main (needs a GPU): records what this rank was given
        seen.append(name); time.sleep(0.05 * (rank + 1)); return 20.0 + len(name)
    out = L.run(["--vid_names", "a", "bb", "ccc", "dddd", "eeeee"], backend="gloo", stage1_main=fake_stage1)
    q.put((rank, seen, out))


def test_multi_video_launcher_shards_and_gathers():
    """launch_videos.py: five videos over two ranks (gloo, CPU): every video runs exactly once on its shard's rank,
    the job time is the MAX over ranks, rank 0 reports all results."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_launcher_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, seen0, out0), (_, seen1, out1) = res
    assert seen0 == ["a", "ccc", "eeeee"] and seen1 == ["bb", "dddd"]
    assert out1 is None and out0["videos"] == 5 and out0["n_gpus"] == 2
    assert out0["psnr"] == {"a": 21.0, "bb": 22.0, "ccc": 23.0, "dddd": 24.0, "eeeee": 25.0}
    assert out0["wall_s"] >= 0.19                    # rank 1: two videos at 0.1 s each


@pytest.mark.parametrize("n", [2, 8])
def test_bare_bench_invocation_spawns_its_own_ranks(n):
    """`python bench.py --gpus N` with NO launcher and no RANK in the environment (how the driver calls it) must not die on
    an assertion: it re-execs itself under torch.distributed.run, all ranks reach init_process_group, the barrier-bracketed
    region and its MAX all-reduce.  AF_BENCH_DRY_RUN=gloo stops short of the GPU so this runs on CPU — at N = 2 and at the
    node's full N = 8 (one video per rank, host threads capped per rank, per-rank times gathered for the JSON line)."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "OMP_NUM_THREADS")}
    env["AF_BENCH_DRY_RUN"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "3", "--warmup", "1"], env=env, cwd="/tmp",
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout                   # ONE JSON line, from rank 0
    out = json.loads(lines[0])
    assert out["dry_run"] and out["n_gpus"] == n and out["video_of_rank0"] == [0]
    assert out["videos"] == [[r_] for r_ in range(n)]   # one independent video per rank
    assert out["max_region_s"] >= 0.01 * n             # rank n-1 sleeps 10 n ms: every rank reports the MAX
    ms = out["ranks"]["ms_per_step_by_rank"]
    assert len(ms) == n and out["ranks"]["devices"] == ["cpu:%d" % r_ for r_ in range(n)]
    assert out["ranks"]["ms_per_step_max"] == max(ms) and abs(max(ms) * 1e-3 - out["max_region_s"]) < 2e-3
    assert int(out["omp_num_threads"]) == max(1, (os.cpu_count() or n) // n)      # host threads capped per rank


def test_gpu_count_check_names_what_it_saw():
    """`--gpus 8` on a box with fewer visible devices must fail with ONE clear line before anything is spawned."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "AF_BENCH_DRY_RUN")}
    try:
        import torch
        if torch.cuda.is_available() and torch.cuda.device_count() >= 64:
            pytest.skip("a box with 64 GPUs")
    except ImportError:
        pass
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "64"], env=env, cwd="/tmp", capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    msg = [l for l in r.stderr.splitlines() if l.startswith("bench.py:")]
    assert len(msg) == 1 and "--gpus 64" in msg[0] and "sees" in msg[0] and "nothing launched" in msg[0], r.stderr[-1000:]
