"""The 256x256 hidden-layer products of the forward / backward chains run either on the fp32 matrix pipe (mode 0,
mlp.hip) or fp32-faithfully on the bf16 matrix pipe (mode 1, the default; mlpbf.hip: weights pre-split into three bf16
images by k_adam, activations split in registers, six partial products per product).  Same nets, same rows: forward
outputs, loss terms and reduced gradients of the two modes must agree to fp32 round-off."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _state(af, two_layer, seed=7):
    import aiod_amd
    import bench
    sds = bench.init_state_dicts(seed, two_layer)
    for net in af.nets:
        af.load_state_dict(net, sds[net])
    af.pre_train_mapping(2, seed=3)
    if two_layer:
        af.pre_train_mapping(2, seed=4, net=aiod_amd.NET_MAPPING2)
    return {net: af.state_dict(net) for net in af.nets}


@pytest.mark.parametrize("two_layer", [False, True])
def test_mlp_modes_agree_at_full_size(two_layer):
    import aiod_amd
    import bench
    dev = torch.device("cuda", 0)
    resx, resy, F = 768, 432, 80
    video = bench.synth_video_device(resx, resy, F, seed=5, device=dev)
    if two_layer:
        video = video + (bench.synth_fg_mask_device(resx, resy, F, seed=5, device=dev),)
    af = aiod_amd.AtlasFit(aiod_amd.default_config(resx, resy, F, two_layer=two_layer))
    af.upload_video(*video)
    sds = _state(af, two_layer)
    g = torch.Generator().manual_seed(13)
    rows = np.zeros((4096, 4), np.float32); rows[:, :3] = torch.rand(4096, 3, generator=g).numpy() * 2 - 1
    fw = {}
    for mode in (0, 1, 3):
        af.set_mlp_mode(mode)
        fw[mode] = {net: af.debug_forward(net, rows) for net in af.nets}
    for net in af.nets:
        d, d3 = np.abs(fw[1][net] - fw[0][net]).max(), np.abs(fw[3][net] - fw[0][net]).max()
        print("two_layer", two_layer, "net", net, "forward max |bf16x6 - fp32 MFMA| %.3g   |f16x3 - fp32 MFMA| %.3g" % (d, d3))
        assert d < 3e-6, (net, d)
        assert d3 < 3e-6, (net, d3)
    for it in (0, 6000):
        inds = torch.randint(F * resx * resy, (af.N,), generator=g).numpy()
        res = {}
        for mode in (0, 1, 2, 3):                        # 2: the forward of mode 1, backward chain on three products (experiment); 3: f16x3 (mlphf.hip)
            af.set_mlp_mode(mode)
            for net in af.nets:
                af.load_state_dict(net, sds[net])
                z = np.zeros(af.param_count(net), np.float32)
                af.set_adam_state(net, z, z, 0)
            af.set_debug(True)
            losses = af.train_steps(it, 1, inds)[0]
            res[mode] = (losses.copy(), {net: af.last_grads(net) for net in af.nets})
        n = 12 if two_layer else 6
        rel = np.abs(res[1][0][:n] - res[0][0][:n]) / np.maximum(np.abs(res[0][0][:n]), 1e-9)
        print("iter", it, "loss terms rel", rel.max())
        assert rel.max() < 2e-5, (it, res[1][0], res[0][0])
        assert np.array_equal(res[2][0], res[1][0])                              # same forward, same loss record
        for net in af.nets:
            g0, g1, g2 = res[0][1][net], res[1][1][net], res[2][1][net]
            r = np.linalg.norm(g1 - g0) / np.linalg.norm(g0)
            r3 = np.linalg.norm(g2 - g1) / np.linalg.norm(g1)
            print("iter", it, "net", net, "gradient rel (L2) bf16x6 vs fp32 MFMA chains %.3g ; three-product backward chain vs bf16x6 %.3g" % (r, r3))
            assert r < 1e-3, (it, net, r)      # forward differences of ~2e-7 in uv are amplified by the finite-difference rigidity terms
            assert r3 < 3e-4, (it, net, r3)    # 16-bit-mantissa operands in dX = W^T dZ
            rh = np.linalg.norm(res[3][1][net] - g0) / np.linalg.norm(g0)
            print("iter", it, "net", net, "gradient rel (L2) f16x3 vs fp32 MFMA chains %.3g" % rh)
            assert rh < 1e-3, (it, net, rh)    # the same bound bf16x6 is held to
        relh = np.abs(res[3][0][:n] - res[0][0][:n]) / np.maximum(np.abs(res[0][0][:n]), 1e-9)
        print("iter", it, "loss terms rel, f16x3 vs fp32 MFMA", relh.max())
        assert relh.max() < 2e-5, (it, res[3][0], res[0][0])
    af.set_mlp_mode(1)
    af.close()
    del video
    torch.cuda.empty_cache()


def test_f16x3_reports_a_weight_beyond_its_images_range():
    """mlphf.hip scales the fp16 weight images by a fixed 2^12 (finite for |w| < 16): k_adam raises the handle's range flag at |w| >= 8 and
    af_train_steps returns AF_ERANGE (include/atlasfit.h) instead of training on with a saturated image; the bf16x6 chains (mode 1) have no such limit
    and go on from the same state."""
    import aiod_amd
    from oracle import atlas_oracle as O
    import bench
    v = O.synthetic_video(64, 48, 4, seed=0)
    af = aiod_amd.AtlasFit(aiod_amd.default_config(64, 48, 4, samples_batch=256))
    af.upload_video(v.video_frames, v.optical_flows, v.optical_flows_reverse, v.optical_flows_mask, v.optical_flows_reverse_mask)
    sds = bench.init_state_dicts(3)
    for net in af.nets:
        af.load_state_dict(net, sds[net])
    assert af.arithmetic["mlp_mode"] == 3                                            # the library's default
    assert np.isfinite(af.train_steps(0, 2, None, seed=1)).all()
    big = {k: np.array(val, copy=True) for k, val in af.state_dict(aiod_amd.NET_MAPPING1).items()}
    big["hidden.2.weight"][5, 7] = 9.0
    af.load_state_dict(aiod_amd.NET_MAPPING1, big)
    with pytest.raises(aiod_amd.AtlasFitError) as e:
        af.train_steps(2, 1, None, seed=1)
    assert e.value.code == -6 and "fp16 weight images" in str(e.value), (e.value.code, str(e.value))
    af.set_mlp_mode(1)                                                               # the cross-check arithmetic takes the same state
    af.load_state_dict(aiod_amd.NET_MAPPING1, big)
    assert np.isfinite(af.train_steps(3, 1, None, seed=1)).all()
    af.load_state_dict(aiod_amd.NET_MAPPING1, sds[aiod_amd.NET_MAPPING1])           # (the flag is sticky until a call reports it: back in range BEFORE the switch re-emits the fp16 images)
    af.set_mlp_mode(3)
    assert np.isfinite(af.train_steps(4, 1, None, seed=1)).all()                     # and mode 3 trains again once the weight is back in range
    # what the stage-1 CLIs do (AtlasFit.range_fallback): the steps of the reporting call are complete and valid (the images are finite below 16), so the
    # run goes on from the same state on the bf16x6 chains instead of stopping, and says so
    af.range_fallback = True
    af.load_state_dict(aiod_amd.NET_MAPPING1, big)
    l3 = af.train_steps(5, 2, None, seed=1)                                          # reports AF_ERANGE inside, returns the losses of both steps
    assert np.isfinite(l3).all() and af.arithmetic["mlp_mode"] == 1 and any("range fallback" in o for o in af.arithmetic["overrides"]), af.arithmetic
    assert np.isfinite(af.get_params_flat(aiod_amd.NET_MAPPING1)).all()
    assert np.isfinite(af.train_steps(7, 2, None, seed=1)).all() and af.arithmetic["mlp_mode"] == 1     # ... and the following calls run in mode 1 without a report
    af.close()
