"""MI355X-native (gfx950) implementation of the stage-1 neural-atlas optimisation loop of
All-In-One-Deflicker (reference: src/stage1_neural_atlas.py).  The compute path is libatlasfit.so
(hand-written HIP, C ABI in include/atlasfit.h); this package is the thin host-side mirror of the
reference's Python interface for that path.  There is NO CPU fallback: without the HIP library and a
GPU every compute entry point raises."""
from .atlasfit import AtlasFit, AtlasFitError, AfConfig, default_config, load_library, NET_MAPPING1, NET_ATLAS, NET_MAPPING2, NET_ALPHA  # noqa: F401
