"""This path's own run-to-run distribution of the configs[1] PSNR at the switch for ONE seed (GPU box; a diagnostic, not product):
python tools/c2_seed_spread.py SEED — the schedule up to iteration 5000 on nine split-K partitions of k_dw (another summation order, nothing else)
and in the three arithmetics on the shipped partition."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import test_gpu_c2 as T   # noqa: E402

PARTS = (None, "306,150,126,129,87", "306,170,145,148,100", "306,160,135,138,93", "306,140,118,120,80", "290,150,126,129,87", "320,150,126,129,87",
         "306,155,130,133,90", "306,165,140,143,96")


def main():
    seed = int(sys.argv[1])
    recs = T._records(); rec = recs[seed]; arms2 = T._second_arms(recs)
    video = T._video(seed, rec)
    out = []
    for part in PARTS:
        p_pre, p_at, _, losses = T._run(seed, rec, part, 5000, video)
        out.append(p_at[5000])
        print("seed %d partition %-20s PSNR after the pre-train %.4f, after 5000 iterations %.4f ; total loss, mean over iterations 3000..4999: %.2f"
              % (seed, part or "shipped", p_pre, p_at[5000], losses[3000:5000, 5].mean()), flush=True)
    os.environ["AF_EXPERIMENT"] = "1"
    for name, m in (("bf16x6 chains + bf16x6 k_dw", "1"), ("fp32-MFMA chains + fp32-MFMA k_dw", "0")):
        os.environ["AF_MLP_MODE"] = m; os.environ["AF_DW_MODE"] = m
        p_pre, p_at, _, losses = T._run(seed, rec, None, 5000, video)
        print("seed %d %-36s PSNR after the pre-train %.4f, after 5000 iterations %.4f ; total loss, mean over iterations 3000..4999: %.2f"
              % (seed, name, p_pre, p_at[5000], losses[3000:5000, 5].mean()), flush=True)
    o = np.array(out)
    refs = [rec["psnr_at"][5000]] + ([arms2[seed][1][5000]] if seed in arms2 else [])
    print("f16x3 over %d partitions: mean %.4f, sd %.4f, min %.4f, max %.4f dB ; reference arms in the fixtures: %s" % (len(o), o.mean(), o.std(ddof=1), o.min(), o.max(), np.round(refs, 4)))


if __name__ == "__main__":
    main()
