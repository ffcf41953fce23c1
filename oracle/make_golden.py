"""Generate tests/golden/*.npz by running the REFERENCE's own modules (read-only import from
/root/reference) on seeded synthetic inputs, and check the oracle restatement (oracle/atlas_oracle.py)
against them while doing so.  Run in the build container only:

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden.py

The fixtures travel with the repo; /root/reference does not exist on the GPU box.
"""
import os
import sys
import types

import numpy as np
import torch

REF = os.environ.get("AF_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.dont_write_bytecode = True
sys.path.insert(0, REF)
sys.path.insert(0, ROOT)
for _name in ("cv2", "imageio"):                       # unwrap_utils imports them at module scope only
    sys.modules.setdefault(_name, types.ModuleType(_name))

from src.models.stage_1.implicit_neural_networks import IMLP                       # noqa: E402
from src.models.stage_1.loss_utils import (get_gradient_loss_single, get_rigidity_loss,   # noqa: E402
                                           get_optical_flow_loss)
from src.models.stage_1.unwrap_utils import get_tuples, pre_train_mapping           # noqa: E402

from oracle import atlas_oracle as O                                                # noqa: E402

CONFIG = {
    "samples_batch": 256, "optical_flow_coeff": 500.0, "derivative_amount": 1, "rgb_coeff": 5000,
    "rigidity_coeff": 1.0, "uv_mapping_scale": 0.8, "number_of_channels_atlas": 256, "number_of_layers_atlas": 8,
    "number_of_channels_mapping1": 256, "number_of_layers_mapping1": 6, "gradient_loss_coeff": 1000,
    "use_gradient_loss": True, "positional_encoding_num_atlas": 10, "use_positional_encoding_mapping1": False,
    "number_of_positional_encoding_mapping1": 4, "include_global_rigidity_loss": True,
    "global_rigidity_derivative_amount_fg": 100, "global_rigidity_coeff_fg": 5.0, "stop_global_rigidity": 5,
}
RESX, RESY, NF, VSEED, WSEED = 40, 24, 6, 3, 1234
K_ITERS = 10           # iterations 0..9: the global-rigidity term switches off after iteration 5
PRE_STEPS_ITERS = 1    # 1 x NF pre-train steps
PRE_BATCH = 512


def ref_models(seed):
    torch.manual_seed(seed)
    m = IMLP(input_dim=3, output_dim=2, hidden_dim=256, use_positional=False, positional_dim=4, num_layers=6, skip_layers=[], verbose=False)
    a = IMLP(input_dim=2, output_dim=3, hidden_dim=256, use_positional=True, positional_dim=10, num_layers=8, skip_layers=[4, 7], verbose=False)
    return m, a


def ref_iteration(i, jif_current, v, mapping, atlas, c, device="cpu"):
    """The loop body of src/stage1_neural_atlas.py:153-227 driven with the reference's own functions."""
    nf, L = v.F, v.larger_dim
    rgb_current = v.video_frames[jif_current[1, :], jif_current[0, :], :, jif_current[2, :]].squeeze(1)
    xyt = torch.cat((jif_current[0, :] / (L / 2) - 1, jif_current[1, :] / (L / 2) - 1, jif_current[2, :] / (nf / 2.0) - 1), dim=1)
    uv = mapping(xyt)
    alpha = torch.ones(jif_current.shape[1], 1)
    rgb = (atlas(uv * 0.5 + 0.5) + 1.0) * 0.5
    grad = get_gradient_loss_single(v.video_frames_dx, v.video_frames_dy, jif_current, mapping, atlas, rgb, device, v.resx, nf)
    rgb_l = (torch.norm(rgb - rgb_current, dim=1) ** 2).mean()
    rig = get_rigidity_loss(jif_current, c["derivative_amount"], L, nf, mapping, uv, device, uv_mapping_scale=c["uv_mapping_scale"])
    glob = c["include_global_rigidity_loss"] and i <= c["stop_global_rigidity"]
    if glob:
        grig = get_rigidity_loss(jif_current, c["global_rigidity_derivative_amount_fg"], L, nf, mapping, uv, device, uv_mapping_scale=c["uv_mapping_scale"])
    flow = get_optical_flow_loss(jif_current, uv, v.optical_flows_reverse, v.optical_flows_reverse_mask, L, nf, mapping,
                                 v.optical_flows, v.optical_flows_mask, c["uv_mapping_scale"], device, use_alpha=True, alpha=alpha)
    if glob:
        loss = c["rigidity_coeff"] * rig + c["global_rigidity_coeff_fg"] * grig + rgb_l * c["rgb_coeff"] + c["optical_flow_coeff"] * flow + grad * c["gradient_loss_coeff"]
    else:
        loss = c["rigidity_coeff"] * rig + rgb_l * c["rgb_coeff"] + c["optical_flow_coeff"] * flow + grad * c["gradient_loss_coeff"]
    return loss, [float(rgb_l), float(grad), float(rig), float(grig) if glob else 0.0, float(flow), float(loss)]


def main(flow="constant"):
    """flow="constant": tests/golden/single_small*.npz (the translating video).  flow="field" (round 4): the same networks, pre-train
    and index stream on the video whose flow differs at every pixel of every frame and whose masks have holes
    (oracle.atlas_oracle.synthetic_video(flow="field")) -> single_field.npz; the pre-train does not see the video, so the trajectory
    starts from single_small_start.npz (asserted)."""
    out_dir = os.path.join(ROOT, "tests", "golden")
    os.makedirs(out_dir, exist_ok=True)
    tag = {"constant": "small", "field": "field"}[flow]
    c = CONFIG
    N = c["samples_batch"]
    video = O.synthetic_video(RESX, RESY, NF, seed=VSEED, flow=flow)

    # ---- 1. networks: reference init == oracle init; forward parity on random rows
    rm, ra = ref_models(WSEED)
    om, oa = O.build_single_atlas_models(c, seed=WSEED)
    for (kn, pr), (_, po) in zip(list(rm.state_dict().items()) + list(ra.state_dict().items()),
                                 list(om.state_dict().items()) + list(oa.state_dict().items())):
        assert torch.equal(pr, po), kn
    g = torch.Generator().manual_seed(7)
    rows_xyt = torch.rand(96, 3, generator=g) * 2 - 1
    rows_uv = torch.rand(96, 2, generator=g)
    with torch.no_grad():
        fwd_map = rm(rows_xyt); fwd_atlas = ra(rows_uv)
        assert torch.equal(fwd_map, om(rows_xyt)) and torch.equal(fwd_atlas, oa(rows_uv))

    # ---- 2. get_tuples
    jif_all = get_tuples(NF, video.video_frames)
    assert torch.equal(jif_all, O.get_tuples(NF, RESY, RESX))

    # ---- 3. pre-train (reference function, reference RNG order) vs oracle with injected draws
    torch.manual_seed(WSEED + 1)
    st = torch.get_rng_state()
    ys, xs = [], []
    for _ in range(PRE_STEPS_ITERS * NF):
        ys.append(torch.randint(RESY, (10000, 1))); xs.append(torch.randint(RESX, (10000, 1)))
    torch.set_rng_state(st)
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
        pre_train_mapping(rm, NF, c["uv_mapping_scale"], resx=RESX, resy=RESY, larger_dim=video.larger_dim, device="cpu", pretrain_iters=PRE_STEPS_ITERS)
    ys_t = torch.stack([y.view(-1) for y in ys]); xs_t = torch.stack([x.view(-1) for x in xs])
    O.pre_train_mapping(om, NF, c["uv_mapping_scale"], RESX, RESY, video.larger_dim, PRE_STEPS_ITERS, ys_t, xs_t)
    for pr, po in zip(rm.parameters(), om.parameters()):
        assert torch.allclose(pr, po, rtol=0, atol=1e-7), "pre-train restatement diverged"
    # a second, smaller pre-train run that the GPU test replays (batch 512 keeps the fixture small)
    rm2, _ = ref_models(WSEED)
    torch.manual_seed(WSEED + 2)
    ys2 = torch.randint(RESY, (2 * NF, PRE_BATCH)); xs2 = torch.randint(RESX, (2 * NF, PRE_BATCH))
    om2, _ = O.build_single_atlas_models(c, seed=WSEED)
    pre_losses = O.pre_train_mapping(om2, NF, c["uv_mapping_scale"], RESX, RESY, video.larger_dim, 2, ys2, xs2, batch=PRE_BATCH)
    pre_params = O.flat_params(om2)

    # ---- 4. main loop: K iterations with the reference functions; oracle must track it
    opt = torch.optim.Adam([{"params": list(rm.parameters())}, {"params": list(ra.parameters())}], lr=0.0001)
    tr = O.SingleAtlasTrainer(c, video, mapping=om, atlas=oa)
    torch.manual_seed(WSEED + 3)
    inds = torch.stack([torch.randint(jif_all.shape[1], (N, 1)).view(-1) for _ in range(K_ITERS)])
    start_map, start_atlas = O.flat_params(rm), O.flat_params(ra)
    losses = []
    grads0 = None
    for i in range(K_ITERS):
        jif_current = jif_all[:, inds[i].view(-1, 1)]
        loss, terms = ref_iteration(i, jif_current, video, rm, ra, c)
        opt.zero_grad(); loss.backward()
        if i == 0:
            grads0 = (O.flat_grads(rm), O.flat_grads(ra))
        opt.step()
        o_terms = tr.step(i, inds[i])
        ref_t = np.array(terms); ora_t = np.array([o_terms[k] for k in ("rgb", "gradient", "rigidity", "global_rigidity", "flow", "total")])
        assert np.allclose(ref_t, ora_t, rtol=2e-5, atol=1e-7), (i, ref_t, ora_t)
        losses.append(terms)
    end_map, end_atlas = O.flat_params(rm), O.flat_params(ra)
    assert np.allclose(end_map, O.flat_params(om), atol=2e-6) and np.allclose(end_atlas, O.flat_params(oa), atol=2e-6)
    ref_psnr, _ = O.mean_psnr(rm, ra, video)

    # ---- 5. portrait aspect (resy > resx): the gradient loss normalises by resx while everything else uses
    # larger_dim = resy (stage1_neural_atlas.py:186-188) — one iteration with and without the global term
    pv = O.synthetic_video(24, 40, 5, seed=VSEED + 1, flow=flow)
    prm, pra = ref_models(WSEED + 7)
    pom, poa = O.build_single_atlas_models(c, seed=WSEED + 7)
    ptr = O.SingleAtlasTrainer(c, pv, mapping=pom, atlas=poa)
    pj = get_tuples(5, pv.video_frames)
    assert torch.equal(pj, O.get_tuples(5, 40, 24))
    pg = torch.Generator().manual_seed(5)
    portrait = []
    for it in (0, K_ITERS + 1):
        pinds = torch.randint(pj.shape[1], (N,), generator=pg)
        _, terms = ref_iteration(it, pj[:, pinds.view(-1, 1)], pv, prm, pra, c)
        o_terms = ptr.loss_and_grads(it, pinds)
        ora_t = np.array([o_terms[k] for k in ("rgb", "gradient", "rigidity", "global_rigidity", "flow", "total")])
        assert np.allclose(np.array(terms), ora_t, rtol=2e-5, atol=1e-7), (it, terms, ora_t)
        portrait.append(terms)

    if flow != "constant":         # same seeds, same pre-train: the state the loop starts from is the constant-flow fixture's
        st0 = np.load(os.path.join(out_dir, "single_small_start.npz"))
        assert np.array_equal(st0["start_map"], start_map) and np.array_equal(st0["start_atlas"], start_atlas)
    if os.environ.get("AF_GOLDEN_CHECK_ONLY"):
        print("restatement == reference modules on the %s-flow video (check only, fixtures untouched)" % flow)
        return
    np.savez_compressed(
        os.path.join(out_dir, "single_%s.npz" % tag),
        resx=RESX, resy=RESY, nframes=NF, video_seed=VSEED, weight_seed=WSEED, samples_batch=N,
        config_keys=np.array(sorted(c.keys())), config_vals=np.array([float(c[k]) for k in sorted(c.keys())]),
        rows_xyt=rows_xyt.numpy(), rows_uv=rows_uv.numpy(), fwd_map=fwd_map.numpy(), fwd_atlas=fwd_atlas.numpy(),
        init_checksum=np.array([float(np.abs(O.flat_params(ref_models(WSEED)[0])).sum()), float(np.abs(O.flat_params(ref_models(WSEED)[1])).sum())]),
        pre_ys=ys2.numpy().astype(np.int16), pre_xs=xs2.numpy().astype(np.int16), pre_batch=PRE_BATCH, pre_losses=np.array(pre_losses, np.float32),
        pre_params_sample=pre_params[::97].copy(),
        start_map_sample=start_map[::97].copy(), start_atlas_sample=start_atlas[::97].copy(),
        start_map_sum=float(np.abs(start_map).sum()), start_atlas_sum=float(np.abs(start_atlas).sum()),
        inds=inds.numpy().astype(np.int32), losses=np.array(losses, np.float64),
        grads0_map_sample=grads0[0][::97].copy(), grads0_atlas_sample=grads0[1][::97].copy(),
        grads0_map_norm=float(np.linalg.norm(grads0[0])), grads0_atlas_norm=float(np.linalg.norm(grads0[1])),
        end_map_sample=end_map[::97].copy(), end_atlas_sample=end_atlas[::97].copy(),
        psnr=ref_psnr,
        video_checksum=float(video.video_frames.double().sum()), mask_checksum=float(video.optical_flows_mask.sum()),
        flow_checksum=float(video.optical_flows.double().abs().sum() + video.optical_flows_reverse.double().abs().sum()),
        portrait_losses=np.array(portrait, np.float64),
    )
    # the state the main loop started from (after the reference's pre-train) is needed bit-exactly by the
    # GPU trajectory test: store it in full (fp32, 2.7 MB)
    if flow == "constant":
        np.savez_compressed(os.path.join(out_dir, "single_small_start.npz"), start_map=start_map, start_atlas=start_atlas)
    print("golden written:", out_dir, tag, "losses[0] =", losses[0], "psnr =", ref_psnr)


if __name__ == "__main__":
    for kind in sys.argv[1:] or ["constant", "field"]:
        main(kind)
