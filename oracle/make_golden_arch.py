"""Architectures other than the shipped one (config keys number_of_layers_*, stage1_neural_atlas.py:112-128, _seg.py:127-161):
forward outputs of the REFERENCE's own `IMLP` for a set of layer counts of every net kind, on seeded rows and seeded
torch-default weights -> tests/golden/arch_variants.npz.  The GPU test (tests/test_gpu_arch.py) re-creates the same weights
from the seed (nn.Linear init in construction order, RNG-only) and holds the HIP chains against these outputs; the oracle
restatement (oracle/atlas_oracle.py:OracleIMLP) is checked against them here as well.

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden_arch.py          (build container only: imports /root/reference read-only)
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
for _name in ("cv2", "imageio"):
    sys.modules.setdefault(_name, types.ModuleType(_name))

from src.models.stage_1.implicit_neural_networks import IMLP      # noqa: E402
from oracle import atlas_oracle as O                                # noqa: E402

# kind -> (input_dim, output_dim, use_positional, positional_dim, skip_layers)      (the constructor calls of the two stage-1 scripts)
KINDS = {"mapping": (3, 2, False, 4, []), "atlas": (2, 3, True, 10, [4, 7]), "alpha": (3, 1, True, 5, []),
         # mapping nets WITH positional encoding (use_positional_encoding_mapping*, number_of_positional_encoding_mapping* = K): PE 3 -> 6K
         "mappingpe1": (3, 2, True, 1, []), "mappingpe2": (3, 2, True, 2, []), "mappingpe3": (3, 2, True, 3, []),
         "mappingpe4": (3, 2, True, 4, []), "mappingpe5": (3, 2, True, 5, []),
         # fewer frequencies than shipped on the atlas / alpha nets (positional_encoding_num_atlas / _alpha)
         "atlaspe6": (2, 3, True, 6, [4, 7]), "atlaspe1": (2, 3, True, 1, [4, 7]), "alphape3": (3, 1, True, 3, [])}
VARIANTS = [("mapping", n) for n in (2, 3, 4, 5, 6, 7, 8)] + [("atlas", n) for n in (2, 3, 4, 5, 6, 7, 8)] + [("alpha", n) for n in (2, 3, 5, 8)] \
    + [("mappingpe1", 4), ("mappingpe2", 4), ("mappingpe3", 5), ("mappingpe4", 6), ("mappingpe4", 3), ("mappingpe5", 2)] \
    + [("atlaspe6", 8), ("atlaspe1", 5), ("alphape3", 8)]
# round 5: hidden widths other than 256 (number_of_channels_*, implicit_neural_networks.py:20,43-51): (kind, layers, hidden_dim) -> key "<kind>_<layers>_w<width>"
WIDTHS = [("mapping", 6, 128), ("mapping", 6, 64), ("mapping", 4, 200), ("mapping", 2, 100), ("mapping", 3, 1), ("atlas", 8, 128), ("atlas", 5, 72), ("alpha", 8, 128),
          ("alpha", 3, 33), ("mappingpe4", 6, 128)]
ROWS = 96


def main():
    out = {}
    for kind, nl, width in [(k, n, 256) for k, n in VARIANTS] + WIDTHS:
        ind, outd, pos, pdim, skips = KINDS[kind]
        seed = 7000 + 10 * nl + len(kind) + (0 if width == 256 else 1000 + width)
        torch.manual_seed(seed)
        ref = IMLP(input_dim=ind, output_dim=outd, hidden_dim=width, use_positional=pos, positional_dim=pdim, num_layers=nl, skip_layers=skips, verbose=False)
        torch.manual_seed(seed)
        ora = O.OracleIMLP(ind, outd, width, pos, pdim, skips, nl)
        for (kn, pr), (_, po) in zip(ref.state_dict().items(), ora.state_dict().items()):
            assert torch.equal(pr, po), (kind, nl, kn)
        g = torch.Generator().manual_seed(seed + 1)
        rows = torch.rand(ROWS, ind, generator=g) * 2 - 1
        with torch.no_grad():
            y = ref(rows)
            assert torch.equal(y, ora(rows)), (kind, nl)
        key = "%s_%d" % (kind, nl) + ("" if width == 256 else "_w%d" % width)
        out[key + "_seed"] = seed
        out[key + "_rows"] = rows.numpy()
        out[key + "_out"] = y.numpy()
        out[key + "_nparams"] = sum(p.numel() for p in ref.parameters())
        print(key, "params", out[key + "_nparams"], "out[0]", y[0].numpy())
    if os.environ.get("AF_GOLDEN_CHECK_ONLY"):
        print("arch variants: restatement == reference IMLP (check only)")
        return
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "arch_variants.npz"), variants=np.array(["%s_%d" % v for v in VARIANTS]),
                        width_variants=np.array(["%s_%d_w%d" % v for v in WIDTHS]), **out)
    print("written tests/golden/arch_variants.npz")


if __name__ == "__main__":
    main()
