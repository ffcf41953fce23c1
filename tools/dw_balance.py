#!/usr/bin/env python
"""How evenly does k_dw's static split-K schedule load the CUs?  Per-workgroup busy time of one launch (s_memrealtime)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch, aiod_amd, bench
two = "--two-layer" in sys.argv
NB = int(sys.argv[sys.argv.index("--batch") + 1]) if "--batch" in sys.argv else 10000
dev = torch.device("cuda", 0)
af = aiod_amd.AtlasFit(aiod_amd.default_config(768, 432, 80, two_layer=two, samples_batch=NB))
video = bench.synth_video_device(768, 432, 80, seed=0, device=dev)
if two:
    video = video + (bench.synth_fg_mask_device(768, 432, 80, seed=0, device=dev),)
af.upload_video(*video)
sds = bench.init_state_dicts(1, two)
for net in af.nets:
    af.load_state_dict(net, sds[net])
af.dw_clocks(True)
for it in (4000, 6000):
    af.train_steps(it, 5, None, seed=0, return_losses=False)
    c = af.dw_clocks(True).astype(np.float64)
    c = c[c[:, 1] > 0]
    t0 = c[:, 0].min()
    end = (c[:, 1] - t0) / 100.0; dur = (c[:, 1] - c[:, 0]) / 100.0          # us
    print("iter %d: %d workgroups | kernel span %.1f us | per-WG busy: mean %.1f  min %.1f  max %.1f  (mean/max = %.3f) | last start %.1f us"
          % (it, len(c), end.max(), dur.mean(), dur.min(), dur.max(), dur.mean() / dur.max(), (c[:, 0].max() - t0) / 100.0))
    order = np.argsort(dur)
    print("   slowest WGs:", [(int(i), round(float(dur[i]), 1)) for i in order[-5:]], " fastest:", [(int(i), round(float(dur[i]), 1)) for i in order[:5]])
af.close()
