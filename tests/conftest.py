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


@pytest.fixture(scope="session")
def golden():
    g = dict(np.load(os.path.join(GOLDEN, "single_small.npz"), allow_pickle=False))
    s = dict(np.load(os.path.join(GOLDEN, "single_small_start.npz")))
    g.update(s)
    g["config"] = {str(k): float(v) for k, v in zip(g["config_keys"], g["config_vals"])}
    for k in ("samples_batch", "derivative_amount", "number_of_channels_atlas", "number_of_layers_atlas",
              "number_of_channels_mapping1", "number_of_layers_mapping1", "positional_encoding_num_atlas",
              "number_of_positional_encoding_mapping1", "global_rigidity_derivative_amount_fg", "stop_global_rigidity"):
        g["config"][k] = int(g["config"][k])
    for k in ("use_gradient_loss", "use_positional_encoding_mapping1", "include_global_rigidity_loss"):
        g["config"][k] = bool(g["config"][k])
    return g


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


@pytest.fixture(scope="session")
def golden_seg():
    """Fixture of the fg/bg dual-atlas path, produced from the reference's modules by oracle/make_golden_seg.py."""
    g = dict(np.load(os.path.join(GOLDEN, "seg_small.npz"), allow_pickle=False))
    g.update(dict(np.load(os.path.join(GOLDEN, "seg_small_start.npz"))))      # the two pre-trained mapping nets the loop starts from
    g["config"] = _typed_config(g["config_keys"], g["config_vals"])
    return g


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


@pytest.fixture(scope="session")
def small_seg_video(golden_seg):
    from oracle import atlas_oracle as O
    v = O.synthetic_seg_video(int(golden_seg["resx"]), int(golden_seg["resy"]), int(golden_seg["nframes"]), seed=int(golden_seg["video_seed"]))
    assert abs(float(v.video_frames.double().sum()) - float(golden_seg["video_checksum"])) < 1e-6
    assert abs(float(v.mask_frames.double().sum()) - float(golden_seg["mask_checksum"])) < 1e-6
    return v


@pytest.fixture(scope="session")
def small_video(golden):
    from oracle import atlas_oracle as O
    v = O.synthetic_video(int(golden["resx"]), int(golden["resy"]), int(golden["nframes"]), seed=int(golden["video_seed"]))
    assert abs(float(v.video_frames.double().sum()) - float(golden["video_checksum"])) < 1e-6
    assert float(v.optical_flows_mask.sum()) == float(golden["mask_checksum"])
    return v
