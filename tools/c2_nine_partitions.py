"""configs[1] complete schedules of this path on NINE split-K partitions of k_dw per seed (nine summation orders, nothing else) against the reference arms
(GPU box; a diagnostic, not product): this side's own distribution per video at the switch and at the end, and the paired difference with this side's
noise averaged down to a third.  python tools/c2_nine_partitions.py > gpurun_out/r6_c2_nine_partitions.txt"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import test_gpu_c2 as T   # noqa: E402
from tools.c2_seed_spread import PARTS   # noqa: E402


def main():
    recs = T._records(); arms2 = T._second_arms(recs)
    seeds = sorted(recs)
    kind = np.array([recs[s]["flow"] for s in seeds])
    d_mid, d_end, sd_mid, sd_end = [], [], [], []
    for seed in seeds:
        rec = recs[seed]
        video = T._video(seed, rec)
        mid, end = [], []
        for part in PARTS:
            _, p_at, p_end, _ = T._run(seed, rec, part, rec["iters"], video)
            mid.append(p_at[5000]); end.append(p_end)
        del video
        mid, end = np.array(mid), np.array(end)
        r_mid = [rec["psnr_at"][5000]] + ([arms2[seed][1][5000]] if seed in arms2 else [])
        r_end = [rec["psnr_end"]] + ([arms2[seed][2]] if seed in arms2 else [])
        print("seed %d (%s flow): after 5000 iterations hip %s mean %.4f sd %.3f / reference %s ; at the end hip %s mean %.4f sd %.3f / reference %s"
              % (seed, rec["flow"], np.array2string(mid, precision=3, max_line_width=300), mid.mean(), mid.std(ddof=1), np.round(r_mid, 3),
                 np.array2string(end, precision=3, max_line_width=300), end.mean(), end.std(ddof=1), np.round(r_end, 3)), flush=True)
        d_mid.append(mid.mean() - np.mean(r_mid)); d_end.append(end.mean() - np.mean(r_end)); sd_mid.append(mid.std(ddof=1)); sd_end.append(end.std(ddof=1))
    for name, dd, sd in (("after 5000 iterations", d_mid, sd_mid), ("at the end", d_end, sd_end)):
        d, sd = np.array(dd), np.array(sd)
        print("hip (mean of %d partitions) - reference (mean of a seed's arms) %s: per seed %s dB ; mean %+.4f dB, standard error over seeds %.4f dB (n = %d)"
              % (len(PARTS), name, np.array2string(d, precision=3), d.mean(), d.std(ddof=1) / np.sqrt(len(d)), len(d)))
        for k in sorted(set(kind)):
            dk = d[kind == k]
            print("   %s-flow videos: mean %+.4f dB, standard error %.4f dB (n = %d) ; this side's run-to-run sigma, pooled: %.3f dB (per video %s)"
                  % (k, dk.mean(), dk.std(ddof=1) / np.sqrt(len(dk)), len(dk), float(np.sqrt(np.mean(sd[kind == k] ** 2))), np.round(sd[kind == k], 3)))


if __name__ == "__main__":
    main()
