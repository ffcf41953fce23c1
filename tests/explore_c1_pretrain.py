"""Not a test (pytest does not collect it): how much the state after configs[0]'s 8000 pre-train steps moves under 1-ulp
perturbations of one initial weight and under the two dW arithmetics — the chaos probe behind tests/test_gpu_c1.py's
iteration-0 tolerance.  Usage (GPU box): python tests/explore_c1_pretrain.py [seed]"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import aiod_amd, bench
from oracle import atlas_oracle as O

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
full = len(sys.argv) > 2 and sys.argv[2] == "full"          # also run the 1001 iterations and print the final PSNR
g = dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "c1_reference.npz")))
resx, resy, F, pre_iters = int(g["resx"]), int(g["resy"]), int(g["nframes"]), int(g["pretrain_iters"])
v = O.synthetic_video(resx, resy, F, seed=seed)
k = list(g["seeds"]).index(seed)
for variant in (sys.argv[3].split(",") if len(sys.argv) > 3 else ["base", "ulp1", "ulp2", "ulp3", "dw_fp32", "pre50"]):
    af = aiod_amd.AtlasFit(aiod_amd.default_config(resx, resy, F))
    af.upload_video(v.video_frames, v.optical_flows, v.optical_flows_reverse, v.optical_flows_mask, v.optical_flows_reverse_mask)
    sds = bench.init_state_dicts(seed)
    if variant.startswith("ulp") or variant.endswith("_ulp"):
        key = sorted(kk for kk in sds[aiod_amd.NET_MAPPING1] if kk.endswith("weight"))[(int(variant[3:]) if variant[3:].isdigit() else 1) % 3]
        w = sds[aiod_amd.NET_MAPPING1][key].view(-1)
        w[7] = float(np.nextafter(np.float32(w[7].item()), np.float32(10.0)))
    if variant in ("dw_fp32", "all_fp32", "all_fp32_ulp"):
        af.set_dw_mode(0)
    if variant in ("mlp_fp32", "all_fp32", "all_fp32_ulp"):
        af.set_mlp_mode(0)
    for net in af.nets:
        af.load_state_dict(net, sds[net])
    N, P = af.N, F * resx * resy
    steps = pre_iters * F
    ys = torch.empty((steps, 10000), dtype=torch.int64); xs = torch.empty((steps, 10000), dtype=torch.int64)
    for s in range(steps):
        ys[s] = torch.randint(resy, (10000, 1)).view(-1)
        xs[s] = torch.randint(resx, (10000, 1)).view(-1)
    n_pre = pre_iters // 2 if variant == "pre50" else pre_iters
    pl = af.pre_train_mapping(n_pre, ys.numpy()[: n_pre * F], xs.numpy()[: n_pre * F], return_losses=True)
    p_pre, _ = af.psnr()
    iters = int(g["iters"]) if full else 1
    inds = torch.stack([torch.randint(P, (N, 1)).view(-1) for _ in range(iters)])
    l = af.train_steps(0, iters, inds.numpy())
    if full:
        print("%-8s seed %d final PSNR %.4f dB (reference %.4f) ; total loss every 200: %s" % (variant, seed, af.psnr()[0], float(g["psnr"][k]), np.array2string(l[::200, 5], precision=1)), flush=True)
    print("%-8s pre-train loss first %.5f last %.6f (mean of last 80 %.6f) ; PSNR %.4f ; iteration-0 terms %s ; reference %s"
          % (variant, pl[0], pl[-1], float(np.mean(pl[-80:])), p_pre, np.array2string(l[0, :6], precision=4), np.array2string(g["curves"][k][0], precision=4)), flush=True)
    af.close()
