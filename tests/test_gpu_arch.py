"""Architectures other than the shipped one: the config's number_of_layers_* keys (stage1_neural_atlas.py:112-128,
stage1_neural_atlas_seg.py:127-161; implicit_neural_networks.py:16-60) select 2..8 layers per net, use_positional_encoding_mapping* / number_of_positional_encoding_mapping* put a Fourier encoding
(3 -> 6K, K = 1..5) in front of a mapping net — the layer loops of the
chains are runtime loops, the atlas net's skip_layers = [4, 7] apply to the layers it has (a skip on the OUTPUT layer when
num_layers is 5 or 8).  Forward outputs of every variant against a fixture written from the reference's own `IMLP`
(oracle/make_golden_arch.py -> tests/golden/arch_variants.npz), in both arithmetics of the chains, and complete training steps
of non-shipped configurations (other depths, mapping nets with positional encoding; single atlas and fg/bg) against the CPU oracle."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "arch_variants.npz")


def _state_dict(seed, shapes):
    """nn.Linear init in construction order under torch.manual_seed(seed): what IMLP.__init__ draws."""
    torch.manual_seed(seed)
    sd = {}
    for i, (o, k) in enumerate(shapes):
        lin = torch.nn.Linear(k, o)
        sd["hidden.%d.weight" % i] = lin.weight.detach(); sd["hidden.%d.bias" % i] = lin.bias.detach()
    return sd


def _variants(g, kind):
    return [int(v.split("_")[1]) for v in g["variants"] if v.split("_")[0] == kind]


@pytest.mark.parametrize("mlp_mode", [1, 0])
def test_forward_of_every_layer_count_matches_reference_imlp(mlp_mode):
    import aiod_amd
    A = aiod_amd.atlasfit
    g = dict(np.load(GOLDEN))
    worst = {}
    for kind, net, key, width in (("mapping", A.NET_MAPPING1, "number_of_layers_mapping1", 3), ("mapping", A.NET_MAPPING2, "number_of_layers_mapping2", 3),
                                  ("atlas", A.NET_ATLAS, "number_of_layers_atlas", 2), ("alpha", A.NET_ALPHA, "number_of_layers_alpha", 3)):
        for nl in _variants(g, kind):
            name = "%s_%d" % (kind, nl)
            cfg = A.default_config(64, 48, 4, {key: nl}, two_layer=True)
            af = aiod_amd.AtlasFit(cfg)
            try:
                af.set_mlp_mode(mlp_mode)
                assert af.param_count(net) == int(g[name + "_nparams"]), (name, af.param_count(net))
                af.load_state_dict(net, _state_dict(int(g[name + "_seed"]), A.imlp_shapes(net, cfg)))
                rows = np.zeros((g[name + "_rows"].shape[0], 4), np.float32); rows[:, :width] = g[name + "_rows"]
                out = af.debug_forward(net, rows)
                want = g[name + "_out"]
                err = float(np.abs(out[:, :want.shape[1]] - want).max())
                worst[(name, net)] = err
                assert err < 5e-6, (name, net, mlp_mode, err)
            finally:
                af.close()
    # mapping nets WITH positional encoding (use_positional_encoding_mapping*: PE 3 -> 6K in front of layer 0, K = 1..5 frequencies)
    for name in [str(v) for v in g["variants"] if str(v).startswith("mappingpe")]:
        K, nl = int(name.split("_")[0][len("mappingpe"):]), int(name.split("_")[1])
        for net, which in ((A.NET_MAPPING1, "1"), (A.NET_MAPPING2, "2")):
            cfg = A.default_config(64, 48, 4, {"use_positional_encoding_mapping" + which: True, "number_of_positional_encoding_mapping" + which: K,
                                                "number_of_layers_mapping" + which: nl}, two_layer=True)
            af = aiod_amd.AtlasFit(cfg)
            try:
                af.set_mlp_mode(mlp_mode)
                assert af.param_count(net) == int(g[name + "_nparams"]), (name, af.param_count(net))
                af.load_state_dict(net, _state_dict(int(g[name + "_seed"]), A.imlp_shapes(net, cfg)))
                rows = np.zeros((g[name + "_rows"].shape[0], 4), np.float32); rows[:, :3] = g[name + "_rows"]
                err = float(np.abs(af.debug_forward(net, rows)[:, :2] - g[name + "_out"]).max())
                worst[(name, net)] = err
                assert err < 5e-6, (name, net, mlp_mode, err)
            finally:
                af.close()
    # fewer frequencies than shipped on the atlas / alpha nets (positional_encoding_num_atlas 1..10, positional_encoding_num_alpha 1..5)
    for name in [str(v) for v in g["variants"] if str(v).startswith(("atlaspe", "alphape"))]:
        kind = name[:5]; K, nl = int(name.split("_")[0][7:]), int(name.split("_")[1])
        net, width = (A.NET_ATLAS, 2) if kind == "atlas" else (A.NET_ALPHA, 3)
        cfg = A.default_config(64, 48, 4, {"positional_encoding_num_" + kind: K, "number_of_layers_" + kind: nl}, two_layer=True)
        af = aiod_amd.AtlasFit(cfg)
        try:
            af.set_mlp_mode(mlp_mode)
            assert af.param_count(net) == int(g[name + "_nparams"]), (name, af.param_count(net))
            af.load_state_dict(net, _state_dict(int(g[name + "_seed"]), A.imlp_shapes(net, cfg)))
            rows = np.zeros((g[name + "_rows"].shape[0], 4), np.float32); rows[:, :width] = g[name + "_rows"]
            want = g[name + "_out"]
            err = float(np.abs(af.debug_forward(net, rows)[:, :want.shape[1]] - want).max())
            worst[(name, net)] = err
            assert err < 5e-6, (name, net, mlp_mode, err)
        finally:
            af.close()
    print("mlp_mode %d: worst forward distance from the reference IMLP over %d variants: %.3g" % (mlp_mode, len(worst), max(worst.values())))


@pytest.mark.parametrize("two_layer,layers", [(False, dict(number_of_layers_mapping1=3, number_of_layers_atlas=5)),
                                              (False, dict(number_of_layers_mapping1=8, number_of_layers_atlas=2)),
                                              (True, dict(number_of_layers_mapping1=4, number_of_layers_mapping2=2, number_of_layers_atlas=6, number_of_layers_alpha=3)),
                                              (False, dict(use_positional_encoding_mapping1=True, number_of_positional_encoding_mapping1=4)),
                                              (False, dict(use_gradient_loss=False)), (True, dict(use_gradient_loss=False)),
                                              (False, dict(positional_encoding_num_atlas=6)), (True, dict(positional_encoding_num_atlas=7, positional_encoding_num_alpha=2)),
                                              (True, dict(use_positional_encoding_mapping1=True, number_of_positional_encoding_mapping1=3, number_of_layers_mapping1=5,
                                                          use_positional_encoding_mapping2=True, number_of_positional_encoding_mapping2=2))])
def test_training_steps_of_non_shipped_architectures_match_oracle(two_layer, layers, golden, golden_seg, small_video, small_seg_video):
    """Three Adam steps of the whole loop (forward, loss stack, backward, dW, Adam) on the fixture video with other layer counts:
    every loss term within 1e-3 of the CPU oracle built from the same config, end weights close, pre_train_mapping included."""
    import aiod_amd
    from oracle import atlas_oracle as O
    A = aiod_amd.atlasfit
    gd = golden_seg if two_layer else golden
    v = small_seg_video if two_layer else small_video
    cfg = dict(gd["config"]); cfg.update(layers)
    c = A.default_config(int(gd["resx"]), int(gd["resy"]), int(gd["nframes"]), cfg, two_layer=two_layer, pretrain_batch=512)
    af = aiod_amd.AtlasFit(c)
    try:
        if two_layer:
            af.upload_video(v.video_frames, v.optical_flows, v.optical_flows_reverse, v.optical_flows_mask, v.optical_flows_reverse_mask, v.mask_frames)
            models = O.build_seg_models(cfg, seed=77)
            nets = (A.NET_MAPPING1, A.NET_MAPPING2, A.NET_ATLAS, A.NET_ALPHA)
        else:
            af.upload_video(v.video_frames, v.optical_flows, v.optical_flows_reverse, v.optical_flows_mask, v.optical_flows_reverse_mask)
            models = O.build_single_atlas_models(cfg, seed=77)
            nets = (A.NET_MAPPING1, A.NET_ATLAS)
        for net, m in zip(nets, models):
            af.load_state_dict(net, m.state_dict())
        # the 16-row pre-train chains with this depth / input stage: the same injected draws on the device and in the oracle's
        # restatement of pre_train_mapping (unwrap_utils.py:176-198), loss per step compared; then the device state goes into the oracle
        F = int(gd["nframes"])
        for pi, net in enumerate(nets[:2] if two_layer else nets[:1]):
            gen0 = torch.Generator().manual_seed(40 + pi)
            ys = torch.randint(v.resy, (F, 512), generator=gen0); xs = torch.randint(v.resx, (F, 512), generator=gen0)
            pl_h = af.pre_train_mapping(1, ys.numpy(), xs.numpy(), net=net, return_losses=True)
            pl_o = O.pre_train_mapping(models[pi], v.F, cfg["uv_mapping_scale"], v.resx, v.resy, v.larger_dim, 1, ys, xs, batch=512)
            assert np.allclose(pl_h, np.array(pl_o), rtol=1e-4), (layers, net, pl_h, pl_o)
        for net, m in zip(nets, models):
            flat, off = af.get_params_flat(net), 0
            with torch.no_grad():
                for p in m.parameters():
                    p.copy_(torch.from_numpy(flat[off:off + p.numel()].reshape(p.shape))); off += p.numel()
            z = np.zeros(af.param_count(net), np.float32)
            af.set_adam_state(net, z, z, 0)
        tr = O.SegAtlasTrainer(cfg, v, models=models) if two_layer else O.SingleAtlasTrainer(cfg, v, mapping=models[0], atlas=models[1])
        N = int(cfg["samples_batch"])
        gen = torch.Generator().manual_seed(5)
        K = 3
        inds = torch.randint(v.F * v.resx * v.resy, (K, N), generator=gen)
        hip = af.train_steps(0, K, inds.numpy())
        names = O.SEG_TERMS if two_layer else ("rgb", "gradient", "rigidity", "global_rigidity", "flow", "total")
        for k in range(K):
            t = tr.step(k, inds[k])
            want = np.array([t[n] for n in names])
            got = hip[k, :len(names)]
            on = np.abs(want) > 0
            rel = np.abs(got[on] - want[on]) / np.abs(want[on])
            print(layers, "iteration", k, "max rel %.3g" % rel.max())
            assert rel.max() < 1e-3, (layers, k, got, want)
        for net, m in zip(nets, models):
            d = np.abs(af.get_params_flat(net) - O.flat_params(m))
            assert d.max() < 1e-3 and d.mean() < 3e-5, (layers, net, d.max(), d.mean())
    finally:
        af.close()


@pytest.mark.parametrize("mlp_mode", [1, 0])
def test_forward_of_narrower_nets_matches_reference_imlp(mlp_mode):
    """number_of_channels_* below 256 (implicit_neural_networks.py:20,43-51; stage1_neural_atlas.py:69-72,115,124): the net runs zero-padded
    inside the 256-wide chains (host.hip NetDesc::hid) and the ABI speaks the narrow net's own state_dict order.  Forward outputs of ten
    (kind, depth, width) variants — widths 1, 33, 64, 72, 100, 128, 200; with skips, with mapping PE, the loop-free two-layer chain —
    against the reference's IMLP of that width, and the parameter round trip through af_set_params / af_get_params."""
    import aiod_amd
    A = aiod_amd.atlasfit
    g = dict(np.load(GOLDEN))
    worst = 0.0
    for name in [str(v) for v in g["width_variants"]]:
        kind, nl, w = name.split("_")[0], int(name.split("_")[1]), int(name.split("_")[2][1:])
        over = {}
        if kind.startswith("mappingpe"):
            nets = ((A.NET_MAPPING1, "mapping1"), (A.NET_MAPPING2, "mapping2")); width_in = 3
            for _, which in nets:
                over.update({"use_positional_encoding_" + which: True, "number_of_positional_encoding_" + which: int(kind[len("mappingpe"):])})
        elif kind == "mapping":
            nets = ((A.NET_MAPPING1, "mapping1"), (A.NET_MAPPING2, "mapping2")); width_in = 3
        elif kind == "atlas":
            nets = ((A.NET_ATLAS, "atlas"),); width_in = 2
        else:
            nets = ((A.NET_ALPHA, "alpha"),); width_in = 3
        for net, which in nets:
            cfg = A.default_config(64, 48, 4, dict(over, **{"number_of_layers_" + which: nl, "number_of_channels_" + which: w}), two_layer=True)
            af = aiod_amd.AtlasFit(cfg)
            try:
                af.set_mlp_mode(mlp_mode)
                assert af.param_count(net) == int(g[name + "_nparams"]), (name, af.param_count(net))
                sd = _state_dict(int(g[name + "_seed"]), A.imlp_shapes(net, cfg))
                af.load_state_dict(net, sd)
                flat = np.concatenate([np.asarray(t).reshape(-1) for t in sd.values()])
                assert np.array_equal(af.get_params_flat(net), flat), name                 # logical order in, logical order out, bit for bit
                rows = np.zeros((g[name + "_rows"].shape[0], 4), np.float32); rows[:, :width_in] = g[name + "_rows"]
                want = g[name + "_out"]
                err = float(np.abs(af.debug_forward(net, rows)[:, :want.shape[1]] - want).max())
                worst = max(worst, err)
                assert err < 5e-6, (name, net, mlp_mode, err)
            finally:
                af.close()
    print("mlp_mode %d: worst forward distance of the narrower nets from the reference IMLP: %.3g" % (mlp_mode, worst))


@pytest.mark.parametrize("two_layer,widths", [(False, dict(number_of_channels_mapping1=128, number_of_channels_atlas=128)),
                                              (False, dict(number_of_channels_mapping1=40, number_of_channels_atlas=200, number_of_layers_mapping1=2)),
                                              (True, dict(number_of_channels_mapping1=64, number_of_channels_mapping2=200, number_of_channels_atlas=128, number_of_channels_alpha=96))])
def test_training_steps_of_narrower_nets_match_oracle(two_layer, widths, golden, golden_seg, small_video, small_seg_video):
    """The whole loop on narrower nets: pre-train losses, three Adam steps (every loss term within 1e-3 of the CPU oracle built with the same
    widths), end weights and Adam moments close — which they can only be if the padding units stay exactly zero through forward, backward,
    the weight-gradient GEMM and Adam."""
    test_training_steps_of_non_shipped_architectures_match_oracle(two_layer, widths, golden, golden_seg, small_video, small_seg_video)


def test_widths_outside_1_to_256_are_rejected():
    import aiod_amd
    A = aiod_amd.atlasfit
    for key, bad in (("number_of_channels_mapping1", 257), ("number_of_channels_atlas", 0), ("number_of_channels_alpha", 512), ("number_of_channels_mapping2", -3)):
        with pytest.raises(aiod_amd.AtlasFitError):
            aiod_amd.AtlasFit(A.default_config(64, 48, 4, {key: bad}, two_layer=True))


def test_layer_counts_outside_2_to_8_are_rejected():
    import aiod_amd
    A = aiod_amd.atlasfit
    for key, bad in (("number_of_layers_mapping1", 1), ("number_of_layers_atlas", 9), ("number_of_layers_alpha", 0)):
        with pytest.raises(aiod_amd.AtlasFitError):
            aiod_amd.AtlasFit(A.default_config(64, 48, 4, {key: bad}, two_layer=True))
