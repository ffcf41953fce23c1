"""k_dw computes the weight gradients on the fp32 matrix pipe (mode 0), with fp32-faithful bf16x6 split operands on the
bf16 matrix pipe (mode 1, the default) or with two bf16 per operand and three products (mode 2, opt-in: narrower than the
reference's fp32; what round 1's review asked to try, kept as a switch under its acceptance rule - gradient error against an
fp64 twin at full size no more than 3x torch-fp32's, asserted in tests/test_gpu_fullsize.py).  The same forward / backward chains feed all three, so the reduced gradients
differ only by the round-off of the contraction over the row batch: they must agree to fp32 round-off in the regime
the loop runs in (mapping nets pre-trained), at the fixture size and at BASELINE's full size,
single and two-layer.  (From an un-pre-trained init — rigidity ~1e3, row terms cancelling to 1e-3 of their size — ANY two
fp32 summation orders differ by ~5e-4, the reference's own fp32 gradient is 1e-3 from fp64 there: not used here.)"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _grads(af, it, inds, sds):
    out = {}
    for mode in (0, 1, 2):
        af.set_dw_mode(mode)
        for net in af.nets:
            af.load_state_dict(net, sds[net])
            z = np.zeros(af.param_count(net), np.float32)
            af.set_adam_state(net, z, z, 0)
        af.set_debug(True)
        losses = af.train_steps(it, 1, inds)[0]
        out[mode] = (losses.copy(), {net: af.last_grads(net) for net in af.nets})
    af.set_dw_mode(1)
    return out


@pytest.mark.parametrize("two_layer", [False, True])
def test_dw_modes_agree_at_full_size(two_layer):
    import aiod_amd
    import bench
    dev = torch.device("cuda", 0)
    resx, resy, F = 768, 432, 80
    video = bench.synth_video_device(resx, resy, F, seed=3, device=dev)
    if two_layer:
        video = video + (bench.synth_fg_mask_device(resx, resy, F, seed=3, device=dev),)
    af = aiod_amd.AtlasFit(aiod_amd.default_config(resx, resy, F, two_layer=two_layer))
    af.upload_video(*video)
    sds = bench.init_state_dicts(99, two_layer)
    for net in af.nets:
        af.load_state_dict(net, sds[net])
    af.pre_train_mapping(2, seed=3)
    if two_layer:
        af.pre_train_mapping(2, seed=4, net=aiod_amd.NET_MAPPING2)
    sds = {net: af.state_dict(net) for net in af.nets}
    g = torch.Generator().manual_seed(31)
    for it in (0, 6000):
        inds = torch.randint(F * resx * resy, (af.N,), generator=g).numpy()
        r = _grads(af, it, inds, sds)
        assert np.array_equal(r[0][0], r[1][0])                          # the loss record does not depend on k_dw
        assert np.array_equal(r[0][0], r[2][0])
        for net in af.nets:
            g0, g1, g2 = r[0][1][net], r[1][1][net], r[2][1][net]
            rel = np.linalg.norm(g1 - g0) / np.linalg.norm(g0)
            rel3 = np.linalg.norm(g2 - g0) / np.linalg.norm(g0)
            print("two_layer", two_layer, "iter", it, "net", net, "gradient rel (L2) against fp32-MFMA: bf16x6 %.3g, bf16x3 %.3g ; max abs %.3g / %.3g (|g|max %.3g)"
                  % (rel, rel3, np.abs(g1 - g0).max(), np.abs(g2 - g0).max(), np.abs(g0).max()))
            assert rel < 3e-4, (it, net, rel)      # two fp32-faithful summation orders over 90 000 partly cancelling rows
            assert rel3 < 3e-4, (it, net, rel3)    # 16-bit-mantissa operands: ~2^-17 per product, averaged over the batch
    af.close()
    del video
    torch.cuda.empty_cache()


def test_dw_modes_agree_on_ragged_small_batches(golden, small_video):
    """samples_batch 250 on an odd-sized video: segments of one and two row tiles, padded last tiles."""
    import aiod_amd
    from oracle import atlas_oracle as O
    cfg = dict(golden["config"]); cfg.update(samples_batch=250)
    v = O.synthetic_video(37, 21, 5, seed=4)
    af = aiod_amd.AtlasFit(aiod_amd.default_config(v.resx, v.resy, v.F, cfg, pretrain_batch=500))
    af.upload_video(v.video_frames, v.optical_flows, v.optical_flows_reverse, v.optical_flows_mask, v.optical_flows_reverse_mask)
    m, a = O.build_single_atlas_models(cfg, seed=2)
    af.load_state_dict(aiod_amd.NET_MAPPING1, m.state_dict()); af.load_state_dict(aiod_amd.NET_ATLAS, a.state_dict())
    af.pre_train_mapping(40, seed=5)
    sds = {net: af.state_dict(net) for net in af.nets}
    off, flat = 0, af.get_params_flat(aiod_amd.NET_MAPPING1)
    with torch.no_grad():
        for p_ in m.parameters():
            p_.copy_(torch.from_numpy(flat[off:off + p_.numel()].reshape(p_.shape))); off += p_.numel()
    g = torch.Generator().manual_seed(1)
    inds = torch.randint(v.F * v.resx * v.resy, (250,), generator=g)
    r = _grads(af, 0, inds.numpy(), sds)
    tr = O.SingleAtlasTrainer(cfg, v, mapping=m, atlas=a)
    tr.loss_and_grads(0, inds)
    for net, mdl in zip(af.nets, (m, a)):
        g0, g1, g2, go = r[0][1][net], r[1][1][net], r[2][1][net], O.flat_grads(mdl)
        print("net", net, "modes rel bf16x6 %.3g, bf16x3 %.3g ; against the oracle: bf16x6 %.3g, bf16x3 %.3g"
              % (np.linalg.norm(g1 - g0) / np.linalg.norm(g0), np.linalg.norm(g2 - g0) / np.linalg.norm(g0),
                 np.linalg.norm(g1 - go) / np.linalg.norm(go), np.linalg.norm(g2 - go) / np.linalg.norm(go)))
        assert np.linalg.norm(g1 - g0) < 3e-4 * np.linalg.norm(g0) and np.linalg.norm(g2 - g0) < 3e-4 * np.linalg.norm(g0)
        assert np.linalg.norm(g1 - go) < 1e-3 * np.linalg.norm(go) and np.linalg.norm(g2 - go) < 1e-3 * np.linalg.norm(go)
    af.close()
