#!/usr/bin/env python
"""Fit k_dw's cost model (host.hip build_sched) to the hardware: per-workgroup busy time of a launch (s_memrealtime, af_debug_dw_clocks)
against the workgroup's segment list (af_debug_dw_schedule):  time_w = sum_segments (tiles * c_tile[shape] + c_seg[shape]).
Least squares over the 256 workgroups of the 9- and 7-segment schedules of the single- and two-layer handles; prints the per-shape
constants in units of the 8x8 tile cost = 256 (the units build_sched uses)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch, aiod_amd, bench
dev = torch.device("cuda", 0)
rows, rhs = [], []
for two in (False, True):
    af = aiod_amd.AtlasFit(aiod_amd.default_config(768, 432, 80, two_layer=two))
    video = bench.synth_video_device(768, 432, 80, seed=0, device=dev)
    if two:
        video = video + (bench.synth_fg_mask_device(768, 432, 80, seed=0, device=dev),)
    af.upload_video(*video)
    sds = bench.init_state_dicts(1, two)
    for net in af.nets:
        af.load_state_dict(net, sds[net])
    af.pre_train_mapping(1, seed=1)
    af.dw_clocks(True)
    for which, it in ((0, 4000), (1, 6000)):
        af.train_steps(it, 6, None, seed=0, return_losses=False)
        c = af.dw_clocks(True).astype(np.float64)
        sch = af.dw_schedule(which)
        dur = (c[:, 1] - c[:, 0]) / 100.0
        for w in range(sch.shape[0]):
            x = np.zeros(10)
            for shp, t0, t1, _ in sch[w]:
                if shp < 0:
                    break
                x[shp] += t1 - t0; x[5 + shp] += 1
            rows.append(x); rhs.append(dur[w])
        print("two_layer", two, "schedule", which, "span mean %.1f max %.1f us" % (dur.mean(), dur.max()))
    af.close()
A, b = np.array(rows), np.array(rhs)
if "--save" in sys.argv:      # the raw system (per workgroup: tiles per shape, segments per shape | busy us) for offline fits
    np.savez(sys.argv[sys.argv.index("--save") + 1], A=A, b=b)
sol, res, rank, sv = np.linalg.lstsq(A, b, rcond=None)
unit = sol[0] / 256.0
names = ("8x8", "8x2", "8x1", "1x8", "1x2")
print("rank", rank, "rms residual %.2f us" % np.sqrt(((A @ sol - b) ** 2).mean()))
for i, n in enumerate(names):
    print("shape %-3s  per-tile %.3f us = %6.1f units   per-segment %.2f us = %6.1f units" % (n, sol[i], sol[i] / unit, sol[5 + i], sol[5 + i] / unit))
