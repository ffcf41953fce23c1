#!/usr/bin/env python
"""Per-layer weight-gradient distance from an fp64 twin of the CPU restatement at full size (80 x 768x432, N = 10 000), two-layer path,
after a short device pre-train: the HIP path in its arithmetic variants next to torch-fp32, on the constant-flow and the field-flow video.
Diagnostic for tests/test_gpu_fullsize.py::test_full_size_seg_iteration_matches_oracle.  Usage (GPU box): python tools/grad_probe.py [it]"""
import os, sys, copy
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch, aiod_amd, bench
from oracle import atlas_oracle as O
it = int(sys.argv[1]) if len(sys.argv) > 1 else 0
# further arguments: cases "flow[:valid=f][:allvalid][:noglobal]" (random masking of the flow masks / masks forced to 1 / global rigidity off)
CASES = sys.argv[2:] or ["constant", "field"]
dev = torch.device("cuda", 0)
resx, resy, F = 768, 432, 80
nets = (aiod_amd.NET_MAPPING1, aiod_amd.NET_MAPPING2, aiod_amd.NET_ATLAS, aiod_amd.NET_ALPHA)
cfg = dict(aiod_amd.atlasfit.REFERENCE_CONFIG)
for case in CASES:
    flow = case.split(":")[0]
    cfg = dict(aiod_amd.atlasfit.REFERENCE_CONFIG)
    video = bench.synth_video_device(resx, resy, F, seed=1, device=dev, flow=flow)
    for opt in case.split(":")[1:]:
        if opt.startswith("valid="):
            gm = torch.Generator(device=dev).manual_seed(97)
            video = video[:3] + tuple(m * (torch.rand(m.shape, device=dev, generator=gm) < float(opt[6:])).float() for m in video[3:5])
        if opt == "allvalid":
            mk, mr = torch.ones_like(video[3]), torch.ones_like(video[4]); mk[:, :, -1] = 0; mr[:, :, 0] = 0
            video = video[:3] + (mk, mr)
        if opt == "noglobal":
            cfg["include_global_rigidity_loss"] = False
    flow = case
    fg = bench.synth_fg_mask_device(resx, resy, F, seed=1, device=dev)
    af = aiod_amd.AtlasFit(aiod_amd.default_config(resx, resy, F, cfg, two_layer=True))
    af.upload_video(*video, fg)
    sds = bench.init_state_dicts(4321, two_layer=True)
    for net in nets:
        af.load_state_dict(net, sds[net])
    af.pre_train_mapping(2, seed=5, net=aiod_amd.NET_MAPPING1)
    af.pre_train_mapping(2, seed=6, net=aiod_amd.NET_MAPPING2)
    frames, flows, flows_rev, mask, mask_rev = [t.cpu() for t in video]
    v = O.SegVideo(frames, flows[..., None], flows_rev[..., None], mask[..., None], mask_rev[..., None], fg.cpu())
    v64 = O.SegVideo(frames.double(), flows[..., None].double(), flows_rev[..., None].double(), mask[..., None], mask_rev[..., None], fg.cpu().double())
    models = O.build_seg_models(cfg, seed=0)
    for net, m in zip(nets, models):
        flat, off = af.get_params_flat(net), 0
        with torch.no_grad():
            for p in m.parameters():
                p.copy_(torch.from_numpy(flat[off:off + p.numel()].reshape(p.shape))); off += p.numel()
    m64 = [copy.deepcopy(m).double() for m in models]
    for m in m64:
        if m.use_positional:
            m.b = m.b.double()
    tr, tr64 = O.SegAtlasTrainer(cfg, v, models=models), O.SegAtlasTrainer(cfg, v64, models=m64)
    N = cfg["samples_batch"]
    gg = torch.Generator().manual_seed(23)
    batches = {"A": torch.randint(F * resx * resy, (N,), generator=gg), "B": torch.randint(F * resx * resy, (N,), generator=gg)}
    for its, bname in ((it, "A"), (6000 if it == 0 else 0, "A"), (it, "B")) if os.environ.get("PROBE_BATCHES") else ((it, "A"),):
        inds = batches[bname]
        tr.loss_and_grads(its, inds)
        torch.set_default_dtype(torch.float64)
        try:
            tr64.loss_and_grads(its, inds)
        finally:
            torch.set_default_dtype(torch.float32)
        go = [O.flat_grads(m) for m in models]; g64 = [O.flat_grads(m) for m in m64]
        # The flow loss is a NORM (loss_utils.py:303-318): its gradient is the unit vector e / |e| of e = M(p_match) - M(p), undefined at e = 0.
        # Where a sample's match lands within ~1e-5 of its own atlas point the direction is decided by round-off: print the smallest |e| of the
        # batch per mapping net (fp64) and the angle between torch-fp32's and fp64's unit vectors on those rows - the seed there is 12.8 / row.
        jif = tr.jif_all[:, inds.view(-1, 1)]
        for mi in (0, 1):
            with torch.no_grad():
                for fwd, (msk, fl) in ((True, (v.optical_flows_mask, v.optical_flows)), (False, (v.optical_flows_reverse_mask, v.optical_flows_reverse))):
                    xyt = torch.cat((jif[0] / (v.larger_dim / 2) - 1, jif[1] / (v.larger_dim / 2) - 1, jif[2] / (v.F / 2.0) - 1), dim=1)
                    uv32 = models[mi](xyt); uv64 = m64[mi](xyt.double())
                    u_f, x_f, rows = O.flow_matches(jif, msk, fl, v.larger_dim, v.F, fwd, uv32)
                    e32 = models[mi](x_f) - u_f
                    e64 = m64[mi](x_f.double()) - uv64[rows]
                    n64 = e64.norm(dim=1)
                    k = torch.argsort(n64)[:4]
                    cosang = (e32[k].double() * e64[k]).sum(1) / (e32[k].double().norm(dim=1) * n64[k] + 1e-300)
                    print("   mapping%d %s matches: smallest |e| (fp64) %s ; angle between torch-fp32's and fp64's direction there (rad) %s ; median |e| %.3g"
                          % (mi + 1, "fwd" if fwd else "bwd", np.array2string(n64[k].numpy(), precision=3), np.array2string(np.arccos(np.clip(cosang.numpy(), -1, 1)), precision=3), float(n64.median())))
        res = {}
        for mlp_mode, dw_mode in ((1, 1),) if len(sys.argv) > 2 else ((1, 1), (0, 0), (1, 0), (0, 1)):
            af.set_mlp_mode(mlp_mode); af.set_dw_mode(dw_mode)
            for net, mdl in zip(nets, models):
                af.load_state_dict(net, mdl.state_dict())
                z = np.zeros(af.param_count(net), np.float32); af.set_adam_state(net, z, z, 0)
            af.set_debug(True)
            af.train_steps(its, 1, inds.numpy())
            res[(mlp_mode, dw_mode)] = [af.last_grads(net) for net in nets]
        for k, net in enumerate(nets):
            print("== %s flow, iteration %d, batch %s, net %d: |g| %.4g ; whole-net distance from fp64: torch-fp32 %.3g | hip (mlp,dw) %s"
                  % (flow, its, bname, net, np.linalg.norm(g64[k]), np.linalg.norm(go[k] - g64[k]) / np.linalg.norm(g64[k]),
                     "  ".join("%s %.3g" % (m, np.linalg.norm(res[m][k] - g64[k]) / np.linalg.norm(g64[k])) for m in res)))
            off = 0
            for li, (o_, k_) in enumerate(aiod_amd.atlasfit.imlp_shapes(net)):
                cnt = o_ * k_
                n_ = np.linalg.norm(g64[k][off:off + cnt]) + 1e-30
                print("   layer %d weight |g| %-8.3g torch-fp32 %-9.3g %s" % (li, n_, np.linalg.norm(go[k][off:off + cnt] - g64[k][off:off + cnt]) / n_,
                      "  ".join("%s %-9.3g" % (m, np.linalg.norm(res[m][k][off:off + cnt] - g64[k][off:off + cnt]) / n_) for m in res)))
                off += cnt + o_
    af.close(); del video, fg; torch.cuda.empty_cache()
