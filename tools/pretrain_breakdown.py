#!/usr/bin/env python
"""Where a pre_train_mapping step goes: HIP-event times per launch class over 10 x F steps on the BASELINE configs[1] geometry."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch, aiod_amd, bench
dev = torch.device("cuda", 0)
F, W, H = 80, 768, 432
video = bench.synth_video_device(W, H, F, seed=0, device=dev)
af = aiod_amd.AtlasFit(aiod_amd.default_config(W, H, F))
af.upload_video(*video)
sds = bench.init_state_dicts(0)
for net in af.nets: af.load_state_dict(net, sds[net])
af.pre_train_mapping(2, seed=1)
import time
torch.cuda.synchronize(); t0 = time.perf_counter(); af.pre_train_mapping(10, seed=2); af.sync(); t1 = time.perf_counter()
print("untimed: %.4f ms/step" % ((t1 - t0) * 1e3 / (10 * F)))
af.set_timing(0xFFFF)
af.pre_train_mapping(10, seed=3)
t = af.timing(reset=True)
print({k: ("%.2f us" % (v[0] / v[1] * 1e3), v[1]) for k, v in t.items() if v[1]})
af.close()
