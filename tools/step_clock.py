#!/usr/bin/env python
"""The clock and the power INSIDE the real training step (DESIGN.md 4 "Clock").

Runs the bench workload (BASELINE configs[1], iterations centred on 5000) for --steps iterations while a thread samples
`rocm-smi --showpower --showclocks` at >= 10 Hz, then reads the per-workgroup stamps of the LAST step's five hot launches
(af_debug_step_clocks: s_memrealtime and s_memtime at workgroup start and end): ticks / (100 MHz span) = the clock each CU's issue
followed while that launch ran inside the step - not in a back-to-back micro-loop of one kernel.  Prints one JSON object.

    python tools/step_clock.py [--steps 8000] [--two-layer]
"""
import argparse
import json
import os
import re
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np      # noqa: E402
import torch            # noqa: E402
import aiod_amd         # noqa: E402
import bench            # noqa: E402


def sampler(stop, out):
    while not stop.is_set():
        t = time.time()
        try:
            r = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True, timeout=5).stdout
        except Exception:
            break
        p = re.search(r"Power \(W\): ([0-9.]+)", r); s = re.search(r"sclk clock level: \S+ \((\d+)Mhz\)", r); m = re.search(r"mclk clock level: \S+ \((\d+)Mhz\)", r)
        out.append((t, float(p.group(1)) if p else None, int(s.group(1)) if s else None, int(m.group(1)) if m else None))
        time.sleep(max(0.0, 0.08 - (time.time() - t)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=8000)
    ap.add_argument("--two-layer", action="store_true")
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
    first = max(0, 5001 - a.steps // 2)
    af.train_steps(first, 50, None, seed=1, return_losses=False)          # warm
    af.step_clocks(True)                                                  # stamps on (one scalar compare per workgroup)
    samples, stop = [], threading.Event()
    th = threading.Thread(target=sampler, args=(stop, samples)); th.start()
    time.sleep(0.5)
    torch.cuda.synchronize(); t0 = time.time()
    af.train_steps(first, a.steps, None, seed=2, return_losses=False)
    torch.cuda.synchronize(); t1 = time.time()
    stop.set(); th.join()
    st = af.step_clocks(True)
    run = [s for s in samples if t0 + 0.3 <= s[0] <= t1]                   # samples taken while the loop ran (0.3 s in: the ramp is over)
    idle = [s for s in samples if s[0] < t0]
    out = {"workload": "BASELINE configs[%d], %d steps from iteration %d" % (4 if a.two_layer else 1, a.steps, first), "ms_per_step": (t1 - t0) / a.steps * 1e3,
           "rocm_smi": {"samples_while_running": len(run), "sample_rate_hz": len(run) / max(t1 - t0 - 0.3, 1e-9),
                        "power_w": {"median": float(np.median([s[1] for s in run if s[1] is not None])) if run else None,
                                    "min": min((s[1] for s in run if s[1] is not None), default=None), "max": max((s[1] for s in run if s[1] is not None), default=None)},
                        "sclk_mhz": {"median": float(np.median([s[2] for s in run if s[2]])) if run else None, "min": min((s[2] for s in run if s[2]), default=None),
                                     "max": max((s[2] for s in run if s[2]), default=None)},
                        "idle_before": {"power_w": idle[-1][1] if idle else None, "sclk_mhz": idle[-1][2] if idle else None},
                        "series_power_w": [s[1] for s in run][:400], "series_sclk_mhz": [s[2] for s in run][:400]},
           "in_kernel_clock_of_the_last_step": {}}
    for name, c in st.items():
        if len(c) == 0:
            continue
        c = c.astype(np.float64)
        c = c[c[:, 0] > c[:, 0].max() - 2.0e5]                             # stamps of THIS step (an earlier, larger launch of the other row regime leaves older ones)
        span_us = (c[:, 2] - c[:, 0]) / 100.0
        ticks = c[:, 3] - c[:, 1]
        ok = span_us > 5.0                                                  # workgroups that returned at once (rows that do not exist this iteration) carry no clock
        out["in_kernel_clock_of_the_last_step"][name] = {
            "workgroups": int(len(c)), "workgroups_with_work": int(ok.sum()),
            "launch_span_us": float((c[:, 2].max() - c[:, 0].min()) / 100.0),
            "workgroup_span_us_mean": float(span_us[ok].mean()), "ticks_per_workgroup_mean": float(ticks[ok].mean()),
            "clock_mhz_mean": float((ticks[ok] / span_us[ok]).mean()), "clock_mhz_min": float((ticks[ok] / span_us[ok]).min()), "clock_mhz_max": float((ticks[ok] / span_us[ok]).max())}
    af.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
