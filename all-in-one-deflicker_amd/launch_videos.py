"""Multi-video stage-1 launcher: independent videos shard one per GPU over the node, one process per GPU, an RCCL
barrier around the job and nothing else between the ranks (BASELINE.json north_star).  The reference
has no such driver: `test.py` runs one video and never forwards `--gpu` (test.py:36-42).

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29511 \\
        all-in-one-deflicker_amd/launch_videos.py --vid_names clipA clipB ... [--root data/test/] [--down 4] [--two_layer]

Rank r processes videos r, r+G, r+2G, ... on GPU LOCAL_RANK through stage1.main(); rank 0 prints one JSON line with the
per-video PSNR and the wall time of the whole job (MAX over ranks)."""
import argparse
import json
import os
import sys
import time

_HERE = os.path.dirname(os.path.abspath(__file__))


def shard(rank, world, n_videos):
    """Videos of rank `rank`: r, r+world, ...  (every video exactly once, at most ceil(n/world) per rank)."""
    return list(range(rank, n_videos, world))


def run(argv=None, backend=None, stage1_main=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--vid_names", nargs="+", required=True)
    ap.add_argument("--config", type=str, default="config_flow_100.json")
    ap.add_argument("--root", type=str, default="data/test/")
    ap.add_argument("--down", type=int, default=None)
    ap.add_argument("--two_layer", action="store_true")
    ap.add_argument("--seed", type=int, default=None)
    ap.add_argument("--concurrent", type=int, default=1, help="videos optimised at the same time on each GPU (own handle and stream each): "
                    "the kernels of one fill the idle tail rounds of the others, a few percent more aggregate throughput at 2 or 3 (tools/two_videos.py)")
    args = ap.parse_args(argv)
    import torch
    import torch.distributed as dist
    rank, world, local = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))
    use_gpu = torch.cuda.is_available()
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if use_gpu:
            torch.cuda.set_device(local)
            dist.init_process_group(backend or "nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend or "gloo")
    if stage1_main is None:
        if __package__ in (None, ""):
            sys.path.insert(0, os.path.dirname(_HERE))
            import aiod_amd  # noqa: F401
            from aiod_amd import stage1 as S
        else:
            from . import stage1 as S
        cfg_path = args.config if os.path.exists(args.config) else os.path.join("src/config", args.config)
        if os.path.exists(cfg_path):
            config = json.load(open(cfg_path))
        else:
            from aiod_amd.atlasfit import REFERENCE_CONFIG
            config = dict(REFERENCE_CONFIG)

        def stage1_main(name):
            a = argparse.Namespace(vid_path=os.path.join(args.root, name), down=args.down if args.down is not None else (1 if args.two_layer else 4),
                                   device_ordinal=local, seed=args.seed, host_loader=False)
            return S.main(config, a, two_layer=args.two_layer)
    mine = shard(rank, world, len(args.vid_names))
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    if args.concurrent > 1 and len(mine) > 1:
        from concurrent.futures import ThreadPoolExecutor      # the library releases the GIL inside its calls (ctypes)
        with ThreadPoolExecutor(max_workers=args.concurrent) as ex:
            results = dict(zip([args.vid_names[i] for i in mine], ex.map(stage1_main, [args.vid_names[i] for i in mine])))
    else:
        results = {args.vid_names[i]: stage1_main(args.vid_names[i]) for i in mine}
    if use_gpu:
        torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    gathered = [results]
    if world > 1:
        dist.barrier()
        t = torch.tensor([dt], dtype=torch.float64, device=torch.device("cuda", local) if use_gpu else None)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        gathered = [None] * world
        dist.all_gather_object(gathered, results)
        dist.destroy_process_group()
    out = None
    if rank == 0:
        merged = {k: v for g in gathered for k, v in g.items()}
        out = {"videos": len(args.vid_names), "n_gpus": world, "wall_s": dt, "videos_per_hour": 3600.0 * len(args.vid_names) / dt, "psnr": merged}
        print(json.dumps(out))
    return out


if __name__ == "__main__":
    run()
