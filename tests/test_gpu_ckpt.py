"""Resume from checkpoints written by the REFERENCE's own objects (tests/golden/ckpt_single.pt, ckpt_seg.pt:
`IMLP.state_dict()` + `torch.optim.Adam.state_dict()` in the dict of evaluate.py:616-622 / :215-232, produced by
oracle/make_golden_ckpt.py).  `stage1.load_checkpoint` must restore parameters, Adam moments and the step count so that
the next two iterations reproduce what the reference's objects computed from the same state."""
import os

import numpy as np
import pytest
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_reference_checkpoint_layout_is_what_the_loader_parses():
    """CPU: keys and the optimizer's param-group order (single: mapping, atlas; two-layer: mapping1, mapping2, alpha, atlas)."""
    import aiod_amd
    from aiod_amd import stage1 as S
    for name, two_layer, groups in (("ckpt_single.pt", False, [12, 16]), ("ckpt_seg.pt", True, [12, 8, 16, 16])):
        ck = torch.load(os.path.join(GOLDEN, name), map_location="cpu", weights_only=False)
        layout = S._ckpt_layout(two_layer)
        assert set(ck.keys()) == {k for k, _ in layout} | {"iteration", "optimizer_all_state_dict"}
        assert [len(g["params"]) for g in ck["optimizer_all_state_dict"]["param_groups"]] == groups
        idx = 0
        for key, net in layout:                                  # state entries follow the groups; shapes match the nets' layers
            for i, (o, k) in enumerate(aiod_amd.atlasfit.imlp_shapes(net)):
                assert tuple(ck[key]["hidden.%d.weight" % i].shape) == (o, k)
                st = ck["optimizer_all_state_dict"]["state"]
                assert tuple(st[idx]["exp_avg"].shape) == (o, k) and tuple(st[idx + 1]["exp_avg"].shape) == (o,)
                assert int(st[idx]["step"]) == 3
                idx += 2


@pytest.mark.gpu
@pytest.mark.parametrize("two_layer", [False, True])
def test_resume_from_reference_written_checkpoint(two_layer, golden, small_video, golden_seg, small_seg_video):
    import aiod_amd
    from aiod_amd import stage1 as S
    exp = dict(np.load(os.path.join(GOLDEN, "ckpt_expect.npz")))
    g, v = (golden_seg, small_seg_video) if two_layer else (golden, small_video)
    af = aiod_amd.AtlasFit(aiod_amd.default_config(v.resx, v.resy, v.F, g["config"], two_layer=two_layer))
    af.upload_video(v.video_frames, v.optical_flows, v.optical_flows_reverse, v.optical_flows_mask, v.optical_flows_reverse_mask,
                    v.mask_frames if two_layer else None)
    start = S.load_checkpoint(af, os.path.join(GOLDEN, "ckpt_seg.pt" if two_layer else "ckpt_single.pt"))
    assert start == int(exp["save_at"])
    for net in af.nets:
        assert af.adam_state(net)[2] == 3                        # three optimizer steps were taken before the save
    tag = "seg" if two_layer else "single"
    inds, want = exp[tag + "_inds"].astype(np.int64), exp[tag + "_losses"]
    got = af.train_steps(start, inds.shape[0], inds)
    n = want.shape[1]
    rel = np.abs(got[:, :n] - want) / np.maximum(np.abs(want), 1e-9)
    print(tag, "loss rel after resume", rel.max(axis=1))
    assert np.allclose(got[0, :n], want[0], rtol=1e-4, atol=1e-7)          # same state, same batch
    assert np.allclose(got[1, :n], want[1], rtol=1e-3, atol=1e-6)          # one Adam step later: needs the restored moments / step count
    order = (aiod_amd.NET_MAPPING1, aiod_amd.NET_MAPPING2, aiod_amd.NET_ATLAS, aiod_amd.NET_ALPHA) if two_layer else (aiod_amd.NET_MAPPING1, aiod_amd.NET_ATLAS)
    ends = np.concatenate([af.get_params_flat(net)[::97] for net in order])
    d = np.abs(ends - exp[tag + "_end"])
    assert d.max() < 5e-4 and d.mean() < 1e-5, (d.max(), d.mean())
    af.close()
