"""bench.py's multi-GPU timing path — `init_process_group("nccl")` (= RCCL on ROCm), `dist.barrier()` on both sides of the
timed window and the float64 MAX all-reduce of the elapsed time on a device tensor — executed on real hardware with a
world of one rank (the only size a 1-GPU box offers; the N = 2,4,8 runs are the driver's).  The path itself has no data
collective: videos shard one per GPU."""
import os
import socket

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_timed_region_over_rccl_world_size_one():
    import torch.distributed as dist
    import aiod_amd
    import bench
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] = os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1, device_id=dev)
    try:
        cfg = aiod_amd.default_config(96, 54, 8)
        af = aiod_amd.AtlasFit(cfg)
        af.upload_video(*bench.synth_video_device(96, 54, 8, seed=0, device=dev))
        sds = bench.init_state_dicts(1)
        for net in af.nets:
            af.load_state_dict(net, sds[net])
        dt = bench.timed_region(lambda: af.train_steps(0, 5, None, seed=0, return_losses=False), torch.cuda.synchronize, dist, dev)
        assert 0.0 < dt < 5.0
        t = torch.tensor([1.5], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        assert float(t.item()) == 1.5
        assert bench.shard_for_rank(0, 1) == [0]
        af.close()
    finally:
        dist.destroy_process_group()
