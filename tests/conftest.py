import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run through gpurun)")


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


def _load_single(tag):
    g = dict(np.load(os.path.join(GOLDEN, "single_%s.npz" % tag), allow_pickle=False))
    s = dict(np.load(os.path.join(GOLDEN, "single_small_start.npz")))       # the pre-train never sees the video: one start state for both videos
    g.update(s)
    g["config"] = {str(k): float(v) for k, v in zip(g["config_keys"], g["config_vals"])}
    for k in ("samples_batch", "derivative_amount", "number_of_channels_atlas", "number_of_layers_atlas",
              "number_of_channels_mapping1", "number_of_layers_mapping1", "positional_encoding_num_atlas",
              "number_of_positional_encoding_mapping1", "global_rigidity_derivative_amount_fg", "stop_global_rigidity"):
        g["config"][k] = int(g["config"][k])
    for k in ("use_gradient_loss", "use_positional_encoding_mapping1", "include_global_rigidity_loss"):
        g["config"][k] = bool(g["config"][k])
    return g


@pytest.fixture(scope="session")
def golden():
    return _load_single("small")


@pytest.fixture(scope="session")
def golden_field():
    """oracle/make_golden.py field: the reference's modules on the video whose flow differs at every pixel of every frame."""
    return _load_single("field")


def _typed_config(keys, vals):
    cfg = {}
    for k, v in zip(keys, vals):
        k = str(k); v = float(v)
        if k.startswith(("use_", "include_")):
            cfg[k] = bool(v)
        elif k.startswith(("number_of_", "positional_encoding_num", "stop_", "global_rigidity_derivative_amount")) or k in ("samples_batch", "derivative_amount"):
            cfg[k] = int(v)
        else:
            cfg[k] = v
    return cfg


def _load_seg(tag):
    g = dict(np.load(os.path.join(GOLDEN, "seg_%s.npz" % tag), allow_pickle=False))
    g.update(dict(np.load(os.path.join(GOLDEN, "seg_small_start.npz"))))      # the two pre-trained mapping nets the loop starts from
    g["config"] = _typed_config(g["config_keys"], g["config_vals"])
    return g


@pytest.fixture(scope="session")
def golden_seg():
    """Fixture of the fg/bg dual-atlas path, produced from the reference's modules by oracle/make_golden_seg.py."""
    return _load_seg("small")


@pytest.fixture(scope="session")
def golden_seg_field():
    return _load_seg("field")


def seg_start_models(golden_seg):
    """The four nets in the state the fixture's trajectory starts from: seeded init (reference construction order),
    mapping1 / mapping2 replaced by the reference-pre-trained parameters stored in seg_small_start.npz."""
    import torch
    from oracle import atlas_oracle as O
    models = O.build_seg_models(golden_seg["config"], seed=int(golden_seg["weight_seed"]))
    for m, flat in zip(models[:2], (golden_seg["start_m1"], golden_seg["start_m2"])):
        off = 0
        with torch.no_grad():
            for p in m.parameters():
                n = p.numel(); p.copy_(torch.from_numpy(flat[off:off + n].reshape(p.shape))); off += n
        assert off == flat.size
    return models


def _flow_checksum(v):
    return float(v.optical_flows.double().abs().sum() + v.optical_flows_reverse.double().abs().sum())


def _seg_video(g, flow):
    from oracle import atlas_oracle as O
    v = O.synthetic_seg_video(int(g["resx"]), int(g["resy"]), int(g["nframes"]), seed=int(g["video_seed"]), flow=flow)
    assert abs(float(v.video_frames.double().sum()) - float(g["video_checksum"])) < 1e-6
    assert abs(float(v.mask_frames.double().sum()) - float(g["mask_checksum"])) < 1e-6
    if "flow_checksum" in g:       # exact flows: sin/cos/exp of the generator may differ in the last ulp between numpy builds, a transposed field would not pass
        assert abs(_flow_checksum(v) / float(g["flow_checksum"]) - 1.0) < 1e-6
        assert float(v.optical_flows_mask.sum() + v.optical_flows_reverse_mask.sum()) == float(g["flow_mask_checksum"])
    return v


def _video(g, flow):
    from oracle import atlas_oracle as O
    v = O.synthetic_video(int(g["resx"]), int(g["resy"]), int(g["nframes"]), seed=int(g["video_seed"]), flow=flow)
    assert abs(float(v.video_frames.double().sum()) - float(g["video_checksum"])) < 1e-6
    assert float(v.optical_flows_mask.sum()) == float(g["mask_checksum"])
    if "flow_checksum" in g:
        assert abs(_flow_checksum(v) / float(g["flow_checksum"]) - 1.0) < 1e-6
    return v


@pytest.fixture(scope="session")
def small_seg_video(golden_seg):
    return _seg_video(golden_seg, "constant")


@pytest.fixture(scope="session")
def small_seg_video_field(golden_seg_field):
    return _seg_video(golden_seg_field, "field")


@pytest.fixture(scope="session")
def small_video(golden):
    return _video(golden, "constant")


@pytest.fixture(scope="session")
def small_video_field(golden_field):
    return _video(golden_field, "field")
