"""CPU tests: the oracle restatement against the golden fixtures that oracle/make_golden.py produced by
running the reference's own modules, and (when /root/reference is mounted) against those modules live."""
import os

import numpy as np
import pytest
import torch

from oracle import atlas_oracle as O


def _load_flat(model, flat):
    off = 0
    with torch.no_grad():
        for p in model.parameters():
            n = p.numel(); p.copy_(torch.from_numpy(flat[off:off + n].reshape(p.shape))); off += n


def test_init_and_forward_match_reference_fixture(golden):
    m, a = O.build_single_atlas_models(golden["config"], seed=int(golden["weight_seed"]))
    assert abs(float(np.abs(O.flat_params(m)).sum()) - float(golden["init_checksum"][0])) < 1e-3
    assert abs(float(np.abs(O.flat_params(a)).sum()) - float(golden["init_checksum"][1])) < 1e-3
    with torch.no_grad():
        assert np.abs(m(torch.from_numpy(golden["rows_xyt"])).numpy() - golden["fwd_map"]).max() < 1e-6
        assert np.abs(a(torch.from_numpy(golden["rows_uv"])).numpy() - golden["fwd_atlas"]).max() < 1e-6


def test_pe_layout():
    x = torch.tensor([[0.25, -0.5]])
    b = torch.tensor([(2 ** j) * np.pi for j in range(3)])
    pe = O.positional_encoding(x, b)[0]
    exp = []
    for k in range(3):
        exp += [np.sin(0.25 * float(b[k])), np.sin(-0.5 * float(b[k])), np.cos(0.25 * float(b[k])), np.cos(-0.5 * float(b[k]))]
    assert np.allclose(pe.numpy(), np.array(exp, np.float32), atol=1e-6)


def test_trajectory_matches_reference_fixture(golden, small_video):
    m, a = O.build_single_atlas_models(golden["config"], seed=int(golden["weight_seed"]))
    _load_flat(m, golden["start_map"]); _load_flat(a, golden["start_atlas"])
    tr = O.SingleAtlasTrainer(golden["config"], small_video, mapping=m, atlas=a)
    inds = torch.from_numpy(golden["inds"].astype(np.int64))
    for i in range(inds.shape[0]):
        t = tr.step(i, inds[i])
        got = np.array([t[k] for k in ("rgb", "gradient", "rigidity", "global_rigidity", "flow", "total")])
        assert np.allclose(got, golden["losses"][i], rtol=1e-4, atol=1e-7), (i, got, golden["losses"][i])
    assert np.abs(O.flat_params(m)[::97] - golden["end_map_sample"]).max() < 1e-5
    mean, _ = O.mean_psnr(m, a, small_video)
    assert abs(mean - float(golden["psnr"])) < 1e-3


def test_pretrain_matches_reference_fixture(golden):
    m, _ = O.build_single_atlas_models(golden["config"], seed=int(golden["weight_seed"]))
    ys = torch.from_numpy(golden["pre_ys"].astype(np.int64)); xs = torch.from_numpy(golden["pre_xs"].astype(np.int64))
    L = max(int(golden["resx"]), int(golden["resy"]))
    losses = O.pre_train_mapping(m, int(golden["nframes"]), golden["config"]["uv_mapping_scale"], int(golden["resx"]),
                                 int(golden["resy"]), np.int64(L), 2, ys, xs, batch=int(golden["pre_batch"]))
    assert np.allclose(losses, golden["pre_losses"], rtol=1e-5)


def test_empty_flow_set_is_nan_like_the_reference(golden, small_video):
    """loss_utils.py:317-320: a batch without valid flow pixels gives mean(empty) = NaN."""
    v = small_video
    m, a = O.build_single_atlas_models(golden["config"], seed=1)
    tr = O.SingleAtlasTrainer(golden["config"], v, mapping=m, atlas=a)
    # last frame has no forward flow, so pick samples only there AND zero the reverse mask
    v2 = O.Video(v.video_frames, v.optical_flows, v.optical_flows_reverse, v.optical_flows_mask, torch.zeros_like(v.optical_flows_reverse_mask))
    tr.video = v2
    p2 = v.resx * v.resy
    inds = torch.arange(64) + (v.F - 1) * p2
    t = tr.loss_and_grads(0, inds)
    assert np.isnan(t["flow"])


@pytest.mark.skipif(not os.path.isdir("/root/reference/src"), reason="reference not mounted (GPU box)")
def test_restatement_against_live_reference_modules():
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1", AF_GOLDEN_CHECK_ONLY="1")
    r = subprocess.run([sys.executable, os.path.join(root, "oracle", "make_golden.py")], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
