import sys; sys.path.insert(0, '/root/repo')
import torch, numpy as np, aiod_amd, bench
dev = torch.device("cuda", 0)
video = bench.synth_video_device(192, 108, 20, seed=0, device=dev)
sds = bench.init_state_dicts(1)
for N in (9363, 9500, 9700, 9850, 10000, 10400):
    af = aiod_amd.AtlasFit(aiod_amd.default_config(192, 108, 20, samples_batch=N))
    af.upload_video(*video)
    af.load_state_dict(aiod_amd.NET_MAPPING1, sds[aiod_amd.NET_MAPPING1]); af.load_state_dict(aiod_amd.NET_ATLAS, sds[aiod_amd.NET_ATLAS])
    af.train_steps(6000, 5, None, seed=0, return_losses=False)
    af.set_timing(0xFFFF)
    af.train_steps(6000, 20, None, seed=0, return_losses=False)
    t = af.timing()
    ntm = (7 * N + 31) // 32; nta = (3 * N + 31) // 32
    t1 = ntm // 1024 * 1024
    print("N=%d map tiles %d (whole %d, rest %d tiles = %d WG) atlas WG %d | " % (N, ntm, t1, ntm - t1, (ntm - t1 + 3) // 4, (nta + 3) // 4)
          + "  ".join("%s %.3f" % (k, t[k][0] / max(t[k][1], 1)) for k in ("fwd_1", "fwd_2", "bwd_1", "bwd_2", "dw")))
    af.close()
