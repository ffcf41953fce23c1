#!/usr/bin/env python
"""Energy budget of the training step at the board's power limit (DESIGN.md 4 "Energy").

The step runs at ~1370 of 1400 W with 740 W idle, so its time is its DYNAMIC ENERGY over (cap - idle): what has to be ranked is joules,
not issue slots.  This tool (GPU box only) samples the board power (amdgpu hwmon `power1_average` / `power1_input` at ~50 Hz, else
`rocm-smi --showpower` at ~8 Hz) while tools/bin/en_* (tools/energy.hip, built by tools/energy_build.sh) run ONE kernel — or an ablated
build of it at equal work — back to back for --secs seconds per phase, and while the real bench loop runs; per phase

    joules per launch = (mean power of the phase, its first 0.8 s dropped  -  idle power) x HIP-event time per launch.

Component energies are differences between builds at equal work; they are not exactly additive (an ablated build is lighter, clocks
higher and sits elsewhere on the V/f curve: every phase's sclk and power are in the JSON), so the table quotes the full kernel, each
ablation and the residue separately.  Output: one JSON object (phases, per-row-tile energies, the step model next to the measured step).

    python tools/energy_budget.py [--secs 3.5] [--steps 6000]
"""
import argparse
import glob
import json
import os
import re
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tools", "bin")


class Sampler:
    """(time, watts, sclk MHz or None) samples of THE GPU THIS PROCESS SEES in a background thread.

    Two sources run side by side: `rocm-smi --showpower --showclocks` (~8 Hz; it lists the visible device — round 4's step_clock.py read 740 W
    idle / 1370 W loaded through it) and, when the HIP device's PCI bus id (printed by tools/energy.hip) resolves to an amdgpu hwmon node,
    that node's power1_average / power1_input at ~50 Hz.  The box has eight boards in sysfs and `card0` is NOT necessarily ours (round 5's
    first pass read another board: ~1280 W whatever ran here), so the hwmon source is only used if it FOLLOWS the load: its reading under the
    bare-MFMA phase must exceed its idle reading by 200 W, else every window falls back to rocm-smi."""

    def __init__(self, pci_bus_id=None):
        self.samples, self.smi, self.stop = [], [], threading.Event()
        self.hwmon, self.freq, self.use_hwmon = None, None, False
        if pci_bus_id:
            for d in glob.glob("/sys/bus/pci/devices/%s/hwmon/hwmon*" % pci_bus_id.lower()):
                for f in ("power1_average", "power1_input"):
                    q = os.path.join(d, f)
                    try:
                        if float(open(q).read()) > 0:
                            self.hwmon = q
                            break
                    except (OSError, ValueError):
                        pass
                if self.hwmon:
                    self.freq = os.path.join(d, "freq1_input") if os.path.exists(os.path.join(d, "freq1_input")) else None
                    break
        self.threads = [threading.Thread(target=self.run_smi, daemon=True)]
        if self.hwmon:
            self.threads.append(threading.Thread(target=self.run_hwmon, daemon=True))
        for t in self.threads:
            t.start()

    @property
    def source(self):
        return self.hwmon if self.use_hwmon else "rocm-smi --showpower --showclocks"

    def run_hwmon(self):
        while not self.stop.is_set():
            t = time.time()
            try:
                w = float(open(self.hwmon).read()) / 1e6
                s = None
                if self.freq:
                    try:
                        s = float(open(self.freq).read()) / 1e6
                    except (OSError, ValueError):
                        s = None
                self.samples.append((t, w, s))
            except Exception:
                pass
            time.sleep(0.02)

    def run_smi(self):
        while not self.stop.is_set():
            t = time.time()
            try:
                r = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True, timeout=5).stdout
                p = re.search(r"Power \(W\): ([0-9.]+)", r); s = re.search(r"sclk clock level: \S+ \((\d+)Mhz\)", r)
                if p:
                    self.smi.append((0.5 * (t + time.time()), float(p.group(1)), float(s.group(1)) if s else None))
            except Exception:
                time.sleep(0.1)
            time.sleep(max(0.0, 0.05 - (time.time() - t)))

    def window(self, t0, t1, skip=0.8, source=None):
        src = self.samples if (self.use_hwmon if source is None else source == "hwmon") else self.smi
        ws = [(w, s) for t, w, s in src if t0 + skip <= t <= t1 - 0.05]
        if not ws:
            return None, None, 0
        sc = [s for _, s in ws if s]
        return sum(w for w, _ in ws) / len(ws), (sum(sc) / len(sc) if sc else None), len(ws)


def run_phases(binary, secs, phases, sampler):
    out = []
    r = subprocess.run([os.path.join(BIN, binary), str(secs)] + phases, capture_output=True, text=True, timeout=60 + 3 * secs * len(phases))
    for line in r.stdout.splitlines():
        if not line.startswith("PHASE "):
            continue
        kv = dict(x.split("=", 1) for x in line.split()[2:])
        d = {"binary": binary, "phase": line.split()[1], "af_abl": int(kv["AF_ABL"]), "dw_abl": int(kv["DW_ABL"]), "t0": float(kv["t0"]), "t1": float(kv["t1"]),
             "launches": int(kv["launches"]), "ms_per_launch": float(kv["ms_per_launch"]), "units_per_launch": float(kv["units_per_launch"]), "unit": kv["unit"]}
        d["smi"] = sampler.window(d["t0"], d["t1"], source="smi")
        d["hwmon"] = sampler.window(d["t0"], d["t1"], source="hwmon")
        out.append(d)
    if r.returncode != 0 or not out:
        out.append({"binary": binary, "error": (r.stdout + r.stderr)[-400:]})
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--secs", type=float, default=3.5)
    ap.add_argument("--steps", type=int, default=6000)
    a = ap.parse_args()
    # the HIP device's PCI bus id, for the hwmon source
    r0 = subprocess.run([os.path.join(BIN, "en_d0"), "0.05", "pci"], capture_output=True, text=True, timeout=120).stdout
    m = re.search(r"PCI (\S+)", r0)
    sampler = Sampler(m.group(1) if m else None)
    time.sleep(1.0)
    res = {"pci_bus_id": m.group(1) if m else None, "hwmon": sampler.hwmon, "phases": []}
    P = res["phases"]
    P += run_phases("en_d0", a.secs, ["idle", "mfma_bf16", "mfma_f32", "hbm_read", "hbm_copy", "idle", "dw_hbm_1", "dw_l2_1", "dw_hbm_2", "dw_l2_2", "dw_hbm_0"], sampler)
    P += run_phases("en_d1", a.secs, ["dw_hbm_1", "dw_l2_1"], sampler)
    P += run_phases("en_d2", a.secs, ["dw_hbm_1", "dw_l2_1"], sampler)
    for n in (0, 1, 2, 32):
        P += run_phases("en_c%d" % n, a.secs, ["fwd_map", "bwd_map", "fwd_atlas", "bwd_atlas"] + (["bw3_map"] if n == 0 else []), sampler)
    P += run_phases("en_d0", a.secs, ["idle"], sampler)
    # does the hwmon node follow THIS GPU's load?
    def of(phase, key):
        return [p[key][0] for p in P if p.get("phase") == phase and p.get(key) and p[key][0]]
    hw_idle, hw_mfma = of("idle", "hwmon"), of("mfma_bf16", "hwmon")
    sampler.use_hwmon = bool(hw_idle and hw_mfma and max(hw_mfma) - min(hw_idle) > 200.0)
    res["power_source"] = sampler.source
    for p in P:
        if "phase" in p:
            p["power_w"], p["sclk_mhz"], p["power_samples"] = p["hwmon"] if sampler.use_hwmon else p["smi"]
    idles = [p["power_w"] for p in P if p.get("phase") == "idle" and p.get("power_w")]
    idle = min(idles) if idles else None
    res["idle_w"] = idle
    for p in P:
        if p.get("power_w") and p.get("launches"):
            p["joules_per_launch"] = (p["power_w"] - idle) * p["ms_per_launch"] * 1e-3
            p["joules_per_unit"] = p["joules_per_launch"] / p["units_per_launch"] if p["units_per_launch"] else None
    # the real step: the bench workload with power sampled (import here: torch start-up must not sit inside the phases above)
    sys.path.insert(0, ROOT)
    import torch
    import aiod_amd
    import bench
    dev = torch.device("cuda", 0)
    af = aiod_amd.AtlasFit(aiod_amd.default_config(768, 432, 80))
    af.upload_video(*bench.synth_video_device(768, 432, 80, seed=0, device=dev))
    sds = bench.init_state_dicts(1234)
    for net in af.nets:
        af.load_state_dict(net, sds[net])
    af.pre_train_mapping(1, seed=0)
    step = {}
    for name, first in (("9 segments (i <= 5000)", 100), ("7 segments (i > 5000)", 5100)):
        af.train_steps(first, 100, None, seed=1, return_losses=False)
        torch.cuda.synchronize(); t0 = time.time()
        af.train_steps(first, a.steps, None, seed=2, return_losses=False)
        torch.cuda.synchronize(); t1 = time.time()
        w, s, n = sampler.window(t0, t1)
        rows, flops = af.step_work(first)
        step[name] = {"ms_per_step": (t1 - t0) / a.steps * 1e3, "power_w": w, "sclk_mhz": s, "power_samples": n,
                      "joules_per_step_above_idle": (w - idle) * (t1 - t0) / a.steps if (w and idle) else None,
                      "rows_mapping": rows[0], "rows_atlas": rows[1], "algorithmic_gflop": flops / 1e9}
        time.sleep(1.0)
    af.close()
    res["step"] = step
    sampler.stop.set()
    print(json.dumps(res))


if __name__ == "__main__":
    main()
