"""Drop-in for the reference's `src/stage1_neural_atlas_seg.py` (CLI :333-369): the fg/bg dual-atlas stage 1.

    python all-in-one-deflicker_amd/stage1_seg.py --vid_name <name> [--config config_flow_100.json] [--root data/test/] [--down 1] [--gpu 0] [--class_name portrait]

Inputs: `<root>/<vid>/*.png|jpg`, `<root>/<vid>_flow/*.npy` (RAFT, src/preprocess_optical_flow.py) and
`<root>/<vid>_seg/*.png|jpg` (src/preprocess_mask_portrait.py / preprocess_mask_rcnn.py); all three preprocessors are
the reference's own and stay on PyTorch-ROCm.  Everything else is stage1.py with two_layer=True."""
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))


def _cli(argv=None):
    from . import stage1
    return stage1._cli(argv, two_layer=True)


if __name__ == "__main__":
    if __package__ in (None, ""):
        sys.path.insert(0, os.path.dirname(_HERE))
        import aiod_amd  # noqa: F401
        from aiod_amd import stage1 as _s
        _s._cli(None, two_layer=True)
    else:
        _cli()
