#!/usr/bin/env python
"""Experiment: V independent videos (handles, streams) optimised CONCURRENTLY on one GPU from V host threads — do the
kernels of one video fill the idle tail rounds of the other?  Prints aggregate sampled points/s vs the single-video rate."""
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch      # noqa: E402
import aiod_amd   # noqa: E402
import bench      # noqa: E402

V = int(sys.argv[1]) if len(sys.argv) > 1 else 2
K = int(sys.argv[2]) if len(sys.argv) > 2 else 200
first = int(sys.argv[3]) if len(sys.argv) > 3 else 4900
dev = torch.device("cuda", 0)
hs = []
for v in range(V):
    af = aiod_amd.AtlasFit(aiod_amd.default_config(768, 432, 80))
    af.upload_video(*bench.synth_video_device(768, 432, 80, seed=v, device=dev))
    sds = bench.init_state_dicts(100 + v)
    for net in af.nets:
        af.load_state_dict(net, sds[net])
    af.pre_train_mapping(1, seed=v)
    af.train_steps(first - 10, 10, None, seed=v, return_losses=False)
    hs.append(af)
torch.cuda.synchronize()


def run(af, v):
    af.train_steps(first, K, None, seed=v, return_losses=False)


t0 = time.perf_counter(); run(hs[0], 0); t1 = time.perf_counter() - t0
ths = [threading.Thread(target=run, args=(hs[v], v)) for v in range(V)]
t0 = time.perf_counter()
for t in ths:
    t.start()
for t in ths:
    t.join()
tv = time.perf_counter() - t0
N = hs[0].N
print("single video: %.0f points/s (%.3f ms/step) | %d videos concurrently: %.0f points/s aggregate (%.3f ms per video-step)  ratio %.3f"
      % (N * K / t1, t1 / K * 1e3, V, V * N * K / tv, tv / K / V * 1e3, (V * N * K / tv) / (N * K / t1)))
