#!/usr/bin/env python
"""Bisects a batch for the sample(s) behind a weight-gradient discrepancy (tools/grad_probe.py): the HIP path's distance from the fp64 twin of the
CPU restatement on mapping1's last hidden layer, for halves of the batch.  Usage (GPU box): python tools/grad_bisect.py"""
import os, sys, copy
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch, aiod_amd, bench
from oracle import atlas_oracle as O
dev = torch.device("cuda", 0)
resx, resy, F = 768, 432, 80
nets = (aiod_amd.NET_MAPPING1, aiod_amd.NET_MAPPING2, aiod_amd.NET_ATLAS, aiod_amd.NET_ALPHA)
cfg = dict(aiod_amd.atlasfit.REFERENCE_CONFIG)
video = bench.synth_video_device(resx, resy, F, seed=1, device=dev, flow="field")
fg = bench.synth_fg_mask_device(resx, resy, F, seed=1, device=dev)
af0 = aiod_amd.AtlasFit(aiod_amd.default_config(resx, resy, F, cfg, two_layer=True))
af0.upload_video(*video, fg)
sds = bench.init_state_dicts(4321, two_layer=True)
for net in nets:
    af0.load_state_dict(net, sds[net])
af0.pre_train_mapping(2, seed=5, net=aiod_amd.NET_MAPPING1)
af0.pre_train_mapping(2, seed=6, net=aiod_amd.NET_MAPPING2)
frames, flows, flows_rev, mask, mask_rev = [t.cpu() for t in video]
v64 = O.SegVideo(frames.double(), flows[..., None].double(), flows_rev[..., None].double(), mask[..., None], mask_rev[..., None], fg.cpu().double())
models = O.build_seg_models(cfg, seed=0)
for net, m in zip(nets, models):
    flat, off = af0.get_params_flat(net), 0
    with torch.no_grad():
        for p in m.parameters():
            p.copy_(torch.from_numpy(flat[off:off + p.numel()].reshape(p.shape))); off += p.numel()
state = [m.state_dict() for m in models]
af0.close()
m64 = [copy.deepcopy(m).double() for m in models]
for m in m64:
    if m.use_positional:
        m.b = m.b.double()
gg = torch.Generator().manual_seed(23)
A = torch.randint(F * resx * resy, (10000,), generator=gg)
shapes = aiod_amd.atlasfit.imlp_shapes(aiod_amd.NET_MAPPING1)
off4 = sum(o * k + o for o, k in shapes[:4]); cnt4 = shapes[4][0] * shapes[4][1]


def err(inds, verbose=False, over=None):
    c = dict(cfg); c["samples_batch"] = int(len(inds)); c.update(over or {})
    tr64 = O.SegAtlasTrainer(c, v64, models=m64)
    torch.set_default_dtype(torch.float64)
    try:
        t64 = tr64.loss_and_grads(6000, inds)
    finally:
        torch.set_default_dtype(torch.float32)
    g64 = O.flat_grads(m64[0])[off4:off4 + cnt4]
    af = aiod_amd.AtlasFit(aiod_amd.default_config(resx, resy, F, c, two_layer=True))
    af.upload_video(*video, fg)
    for net, sd in zip(nets, state):
        af.load_state_dict(net, sd)
        z = np.zeros(af.param_count(net), np.float32); af.set_adam_state(net, z, z, 0)
    af.set_debug(True)
    hl = af.train_steps(6000, 1, inds.numpy())[0]
    if verbose:
        print("   oracle fp64 terms", np.array([t64[n] for n in O.SEG_TERMS]))
        print("   hip terms        ", hl[:12])
        for k_, net in enumerate(nets):
            gfull, g64f = af.last_grads(net), O.flat_grads(m64[k_])
            o_ = 0
            for li, (oo, kk) in enumerate(aiod_amd.atlasfit.imlp_shapes(net)):
                n_ = np.linalg.norm(g64f[o_:o_ + oo * kk]) + 1e-30
                print("   net %d layer %d weight |g64| %-9.3g hip-vs-fp64 %.3g" % (net, li, n_, np.linalg.norm(gfull[o_:o_ + oo * kk] - g64f[o_:o_ + oo * kk]) / n_))
                o_ += oo * kk + oo
    gh = af.last_grads(aiod_amd.NET_MAPPING1)[off4:off4 + cnt4]
    af.close()
    return float(np.linalg.norm(gh - g64) / np.linalg.norm(g64)), float(np.linalg.norm(gh - g64))


bad = torch.tensor([19622409])
k = int(bad[0]); P2 = resx * resy
f, rem = k // P2, k % P2
y, x = rem // resx, rem % resx
fl = flows[y, x, :, f].numpy(); fr = flows_rev[y, x, :, f].numpy()
hm = np.float32(384.0); hf = np.float32(40.0)
def coords(xx, yy, ff):
    return [np.float32(np.float32(xx) / hm - np.float32(1)), np.float32(np.float32(yy) / hm - np.float32(1)), np.float32(np.float32(ff) / hf - np.float32(1)), np.float32(0)]
rows = np.array([coords(x, y, f), coords(np.float32(x) + fl[0], np.float32(y) + fl[1], f + 1), coords(np.float32(x) + fr[0], np.float32(y) + fr[1], f - 1)], np.float32)
# pre-activations of every hidden layer for the sample's rows (all nine row kinds): is a unit sitting on its ReLU kink?
def nine(xx, yy, ff):
    r = [coords(xx, yy, ff), coords(xx, yy + 1, ff), coords(xx + 1, yy, ff), coords(xx, yy - 1, ff), coords(xx - 1, yy, ff),
         coords(np.float32(xx) + fl[0], np.float32(yy) + fl[1], ff + 1), coords(np.float32(xx) + fr[0], np.float32(yy) + fr[1], ff - 1)]
    return np.array(r, np.float32)
R = nine(x, y, f)
names = ("centre", "y+1", "x+1", "y-1", "x-1", "fwd match", "bwd match")
for mi in (0, 1):
    h32 = torch.from_numpy(R[:, :3]); h64 = h32.double()
    for li, (l32, l64) in enumerate(zip(models[mi].hidden, m64[mi].hidden)):
        with torch.no_grad():
            if li > 0:
                h32 = torch.relu(h32); h64 = torch.relu(h64)
            h32 = l32(h32); h64 = l64(h64)
        if li == len(models[mi].hidden) - 1:
            break
        a64 = h64.abs()
        small = torch.nonzero(a64 < 3e-6)
        for r_, u_ in small.tolist():
            print("mapping%d layer %d row '%s' unit %d: pre-activation fp64 %.3e, torch-fp32 %.3e  <- on the ReLU kink" % (mi + 1, li, names[r_], u_, float(h64[r_, u_]), float(h32[r_, u_])))
        print("mapping%d layer %d: smallest |pre-activation| over the sample's 7 rows x 256 units: %.3e" % (mi + 1, li, float(a64.min())))
