#!/usr/bin/env python
"""The drop-in CLI on a full-size clip, wall clock included: writes a synthetic 80-frame 768x432 clip (PNG frames + RAFT-format
.npy flows [+ masks]) to a scratch folder, then runs all-in-one-deflicker_amd/stage1.py (or stage1_seg.py) on it exactly
as the reference's test.py would (test.py:36-40), shipped config."""
import json, os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
from PIL import Image
import bench
two = "--two-layer" in sys.argv
F, W, H = 80, 768, 432
d = tempfile.mkdtemp(prefix="af_cli_")
dev = torch.device("cuda", 0)
frames, flows, flows_rev, _, _ = bench.synth_video_device(W, H, F, seed=0, device=dev)
os.makedirs(os.path.join(d, "data", "clip")); os.makedirs(os.path.join(d, "data", "clip_flow"))
names = ["%05d.png" % f for f in range(F)]
fr = (frames.permute(3, 0, 1, 2).clamp(0, 1) * 255).round().byte().cpu().numpy()
for f in range(F):
    Image.fromarray(fr[f]).save(os.path.join(d, "data", "clip", names[f]))
fl, flr = flows.permute(3, 0, 1, 2).cpu().numpy(), flows_rev.permute(3, 0, 1, 2).cpu().numpy()
for f in range(F - 1):
    np.save(os.path.join(d, "data", "clip_flow", "%s_%s.npy" % (names[f], names[f + 1])), fl[f])
    np.save(os.path.join(d, "data", "clip_flow", "%s_%s.npy" % (names[f + 1], names[f])), flr[f + 1])
if two:
    os.makedirs(os.path.join(d, "data", "clip_seg"))
    m = (bench.synth_fg_mask_device(W, H, F, seed=0, device=dev).permute(2, 0, 1) * 255).round().byte().cpu().numpy()
    for f in range(F):
        Image.fromarray(m[f]).save(os.path.join(d, "data", "clip_seg", names[f]))
del frames, flows, flows_rev; torch.cuda.empty_cache()
script = os.path.join(ROOT, "all-in-one-deflicker_amd", "stage1_seg.py" if two else "stage1.py")
t0 = time.perf_counter()
r = subprocess.run([sys.executable, script, "--vid_name", "clip", "--root", os.path.join(d, "data"), "--down", "1", "--seed", "1"], cwd=d, capture_output=True, text=True,
                   env=dict(os.environ, AF_CLI_TIMING="1"))
dt = time.perf_counter() - t0
stages = [json.loads(l.split(" ", 1)[1]) for l in r.stderr.splitlines() if l.startswith("AF_CLI_TIMING ")]
res = os.path.join(d, "results", "clip", "stage_1")
psnr = [n for n in os.listdir(os.path.join(res, "010000")) if n.startswith("PSNR_")] if r.returncode == 0 else []
print(json.dumps({"cli": os.path.basename(script), "returncode": r.returncode, "wall_s": dt, "outputs": len(os.listdir(os.path.join(res, "output"))) if r.returncode == 0 else 0,
                  "psnr_marker": psnr, "stage_seconds_inside_main": stages[0] if stages else None, "interpreter_start_and_imports_s": (dt - sum(stages[0].values())) if stages else None, "stderr_tail": r.stderr[-300:] if r.returncode else ""}))
