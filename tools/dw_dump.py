import os, sys
sys.path.insert(0, '/root/repo')
import numpy as np, torch, aiod_amd, bench
dev = torch.device("cuda", 0)
af = aiod_amd.AtlasFit(aiod_amd.default_config(768, 432, 80))
video = bench.synth_video_device(768, 432, 80, seed=0, device=dev)
af.upload_video(*video)
sds = bench.init_state_dicts(1)
for net in af.nets: af.load_state_dict(net, sds[net])
af.dw_clocks(True)
af.train_steps(4000, 6, None, seed=0, return_losses=False)
c = af.dw_clocks(True).astype(np.float64); sch = af.dw_schedule(0)
dur = (c[:, 1] - c[:, 0]) / 100.0
for w in list(range(0, 256, 32)) + list(range(224, 256, 2)):
    segs = [(int(s[0]), int(s[2] - s[1])) for s in sch[w] if s[0] >= 0]
    print(w, round(dur[w], 1), segs)
