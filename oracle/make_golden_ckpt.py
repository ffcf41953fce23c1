"""Checkpoints WRITTEN BY THE REFERENCE's objects, for the resume path (stage1_neural_atlas.py:141-146,
stage1_neural_atlas_seg.py:180-187): the reference's `IMLP`s and a `torch.optim.Adam` in the reference's param-group
order take three loop iterations (the reference's loss functions, as in oracle/make_golden*.py), then the dict of
evaluate.py:616-622 (single atlas) / :215-232 (four nets) is `torch.save`d exactly as the reference does.  The same
objects then run two more iterations from that state on recorded indices; the GPU test loads the file with
`stage1.load_checkpoint`, continues, and must reproduce those losses (the second one only if the Adam moments and the
step count were restored correctly).

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden_ckpt.py        (build container only)
        -> tests/golden/ckpt_single.pt, ckpt_seg.pt (the reference-format files) + ckpt_expect.npz
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.dont_write_bytecode = True
sys.path.insert(0, ROOT)

from oracle import make_golden as G1            # noqa: E402  (imports the reference's modules, read-only)
from oracle import make_golden_seg as G2        # noqa: E402
from oracle import atlas_oracle as O            # noqa: E402
from src.models.stage_1.unwrap_utils import get_tuples   # noqa: E402

SAVE_AT, EXTRA = 2, 2       # evaluate (and save) at iteration 2, then two more iterations from the saved state


def run(two_layer, out_dir):
    if two_layer:
        c, video = G2.CONFIG, O.synthetic_seg_video(G2.RESX, G2.RESY, G2.NF, seed=G2.VSEED)
        models = G2.ref_models(G2.WSEED)                       # mapping1, mapping2, atlas, alpha
        groups = [models[0], models[1], models[3], models[2]]  # stage1_neural_atlas_seg.py:165-169: mapping1, mapping2, alpha, atlas
        step = lambda i, jif: G2.ref_seg_iteration(i, jif, video, *models, c)
        nf = G2.NF
    else:
        c, video = G1.CONFIG, O.synthetic_video(G1.RESX, G1.RESY, G1.NF, seed=G1.VSEED)
        models = G1.ref_models(G1.WSEED)                       # mapping, atlas
        groups = list(models)
        step = lambda i, jif: G1.ref_iteration(i, jif, video, *models, c)
        nf = G1.NF
    opt = torch.optim.Adam([{"params": list(m.parameters())} for m in groups], lr=0.0001)
    jif_all = get_tuples(nf, video.video_frames)
    torch.manual_seed(777 + int(two_layer))
    inds = torch.stack([torch.randint(jif_all.shape[1], (c["samples_batch"], 1)).view(-1) for _ in range(SAVE_AT + 1 + EXTRA)])
    losses = []
    for k in range(SAVE_AT + 1 + EXTRA):
        i = k if k <= SAVE_AT else SAVE_AT + (k - SAVE_AT - 1)          # after the save the loop restarts AT the saved iteration
        loss, terms = step(i, jif_all[:, inds[k].view(-1, 1)])
        opt.zero_grad(); loss.backward(); opt.step()
        losses.append(terms)
        if k == SAVE_AT:
            if two_layer:                                       # evaluate.py:215-222
                torch.save({'F_atlas_state_dict': models[2].state_dict(), 'iteration': i,
                            'model_F_mapping1_state_dict': models[0].state_dict(), 'model_F_mapping2_state_dict': models[1].state_dict(),
                            'model_F_alpha_state_dict': models[3].state_dict(), 'optimizer_all_state_dict': opt.state_dict()},
                           os.path.join(out_dir, "ckpt_seg.pt"))
            else:                                               # evaluate.py:616-622
                torch.save({'F_atlas_state_dict': models[1].state_dict(), 'iteration': i,
                            'model_F_mapping1_state_dict': models[0].state_dict(), 'optimizer_all_state_dict': opt.state_dict()},
                           os.path.join(out_dir, "ckpt_single.pt"))
    return inds[SAVE_AT + 1:].numpy().astype(np.int32), np.array(losses[SAVE_AT + 1:], np.float64), [O.flat_params(m)[::97] for m in models]


def main():
    out_dir = os.path.join(ROOT, "tests", "golden")
    i1, l1, e1 = run(False, out_dir)
    i2, l2, e2 = run(True, out_dir)
    np.savez_compressed(os.path.join(out_dir, "ckpt_expect.npz"), save_at=SAVE_AT, single_inds=i1, single_losses=l1, single_end=np.concatenate(e1),
                        seg_inds=i2, seg_losses=l2, seg_end=np.concatenate(e2))
    print("written; single losses after resume", l1[:, -1], "seg", l2[:, -1])


if __name__ == "__main__":
    main()
