#!/usr/bin/env python
"""How the length of the timed window and of the warm-up move the measured step time (the driver's `--steps 20
--warmup 5` line reads ~3 % below the 200-step line of the same build).

One process, one handle, the bench workload (BASELINE configs[1], iterations centred on 5000).  For every (idle seconds before, warm-up
steps W, timed steps K) it does what bench.py's timed region does - W untimed steps, synchronize, K steps, synchronize - and prints the
ms per step; then a step-by-step timeline right after an idle period (one synchronize per step: +~20 us each, the SHAPE is what counts).

    python tools/window_probe.py [--two-layer]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np      # noqa: E402
import torch            # noqa: E402
import aiod_amd         # noqa: E402
import bench            # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--two-layer", action="store_true")
    ap.add_argument("--reps", type=int, default=3)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    af = aiod_amd.AtlasFit(aiod_amd.default_config(768, 432, 80, two_layer=a.two_layer))
    video = bench.synth_video_device(768, 432, 80, seed=0, device=dev)
    if a.two_layer:
        video = video + (bench.synth_fg_mask_device(768, 432, 80, seed=0, device=dev),)
    af.upload_video(*video)
    sds = bench.init_state_dicts(1234, a.two_layer)
    for net in af.nets:
        af.load_state_dict(net, sds[net])
    af.pre_train_mapping(1, seed=0)

    def window(idle, W, K):
        torch.cuda.synchronize()
        if idle > 0:
            time.sleep(idle)
        first = max(0, 5001 - K // 2)
        if W > 0:
            af.train_steps(max(0, first - W), W, None, seed=1, return_losses=False)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        af.train_steps(first, K, None, seed=2, return_losses=True)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / K * 1e3

    out = {"workload": "BASELINE configs[%d]" % (4 if a.two_layer else 1), "windows": [], "timeline_after_2s_idle_ms": None}
    window(0, 50, 50)
    for idle, W, K in [(0, 5, 20), (2.0, 5, 20), (0, 50, 20), (0, 500, 20), (2.0, 500, 20), (0, 0, 40), (0, 5, 200), (0, 20, 200), (2.0, 20, 200), (0, 20, 2000)]:
        ms = [window(idle, W, K) for _ in range(a.reps)]
        out["windows"].append({"idle_s_before": idle, "warmup_steps": W, "timed_steps": K, "ms_per_step": [round(m, 4) for m in ms], "median": round(float(np.median(ms)), 4)})
        print("idle %.1f s, warm-up %4d, timed %5d steps: %s ms/step" % (idle, W, K, " ".join("%.4f" % m for m in ms)), flush=True)
    torch.cuda.synchronize(); time.sleep(2.0)
    tl = []
    for i in range(120):                      # every step with its own synchronize, from cold
        t0 = time.perf_counter()
        af.train_steps(4941 + i, 1, None, seed=3, return_losses=False)
        torch.cuda.synchronize()
        tl.append((time.perf_counter() - t0) * 1e3)
    out["timeline_after_2s_idle_ms"] = [round(t, 4) for t in tl]
    print("step by step after 2 s idle (one synchronize per step): first 10 %s ; steps 10-19 mean %.4f ; 50-59 mean %.4f ; 110-119 mean %.4f"
          % (" ".join("%.3f" % t for t in tl[:10]), np.mean(tl[10:20]), np.mean(tl[50:60]), np.mean(tl[110:120])), flush=True)
    print(json.dumps(out))
    af.close()


if __name__ == "__main__":
    main()
