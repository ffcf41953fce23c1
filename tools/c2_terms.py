"""Per-term loss curves of configs[1] complete schedules, this path against the reference arms (GPU box; a diagnostic, not product):
python tools/c2_terms.py SEED [SEED ...] — the six terms every 250 iterations as ratios hip / reference (first arm), next to the reference's
second arm / first arm where tests/golden/c2_reference_rerun.npz has one."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import test_gpu_c2 as T   # noqa: E402


def main():
    recs = T._records()
    g2 = dict(np.load(os.path.join(T.GOLD, "c2_reference_rerun.npz")))
    arm2 = {int(s): g2["curves"][k] for k, s in enumerate(g2["seeds"])}
    np.set_printoptions(linewidth=250, precision=3, suppress=True)
    for seed in [int(a) for a in sys.argv[1:]]:
        rec = recs[seed]
        video = T._video(seed, rec)
        for part in T.ALL_PARTITIONS:
            p_pre, p_at, p_end, losses = T._run(seed, rec, part, rec["iters"], video)
            cur = losses[::rec["every"]][:len(rec["curve"]), :6]
            # block means of every term over the 250 iterations after each logged one: the single-iteration values carry the batch's noise
            n = (len(losses) // rec["every"]) * rec["every"]
            blk = losses[:n, :6].reshape(-1, rec["every"], 6).mean(axis=1)
            print("seed %d (%s flow) partition %s: PSNR %.4f -> %s -> %.4f" % (seed, rec["flow"], part or "shipped", p_pre, {k: round(v, 4) for k, v in p_at.items()}, p_end))
            with np.errstate(divide="ignore", invalid="ignore"):
                for t, name in enumerate(T.TERMS):
                    print("   %-16s hip / reference at the logged iterations: %s" % (name, np.nan_to_num(cur[:, t] / rec["curve"][:len(cur), t])))
                    if seed in arm2:
                        print("   %-16s reference arm 2 / arm 1:                  %s" % (name, np.nan_to_num(arm2[seed][:len(cur), t] / rec["curve"][:len(cur), t])))
                print("   hip, means over blocks of %d iterations (rgb, gradient, rigidity, global rigidity, flow, total):" % rec["every"])
                print(blk[::4])


if __name__ == "__main__":
    main()
