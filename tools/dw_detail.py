"""Per-workgroup busy time of one k_dw launch next to the workgroup's segment list and the cost the schedule assigned to it
(tools/dw_balance.py gives the summary).  Usage (GPU box): python tools/dw_detail.py"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch, aiod_amd, bench
dev = torch.device("cuda", 0)
af = aiod_amd.AtlasFit(aiod_amd.default_config(768, 432, 80))
video = bench.synth_video_device(768, 432, 80, seed=0, device=dev)
af.upload_video(*video)
sds = bench.init_state_dicts(1, False)
for net in af.nets: af.load_state_dict(net, sds[net])
af.dw_clocks(True)
af.train_steps(4000, 6, None, seed=0, return_losses=False)
c = af.dw_clocks(True).astype(np.float64)
sch = af.dw_schedule(0)
dur = (c[:, 1] - c[:, 0]) / 100.0
names = ["8x8", "8x2", "8x1", "1x8", "1x2"]
for w in range(0, 256, 1):
    segs = [(names[s[0]], int(s[3]), int(s[1]), int(s[2])) for s in sch[w] if s[0] >= 0]
    cost = sum({"8x8": 306, "8x2": 126, "8x1": 91, "1x8": 91, "1x2": 56}[n] * (t1 - t0) for n, j, t0, t1 in segs) + 60 * len(segs)
    print("wg %3d  %.1f us  model %6d  us/unit %.5f  %s" % (w, dur[w], cost, dur[w] / cost, segs))
