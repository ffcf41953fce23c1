"""configs[1] complete schedules of this path in every arithmetic of the chains, on the seeds of tests/golden/c2_reference.npz (GPU box; tools, not product):
does the arithmetic move where 18 000 Adam steps end?  Per seed one run each with the chains on f16x3 (the default), bf16x6 and fp32 MFMA (k_dw: six bf16
products / six / fp32), same video, initial weights and draws (tests/test_gpu_c2.py's replay), against the mean of the seed's reference arms.
Usage: python tools/c2_modes.py > gpurun_out/r6_c2_arithmetics.txt"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import test_gpu_c2 as T   # noqa: E402

ARITH = (("f16x3 chains, bf16x6 k_dw (default)", None, None), ("bf16x6 chains, bf16x6 k_dw", "1", "1"), ("fp32-MFMA chains, fp32-MFMA k_dw", "0", "0"))


def main():
    recs = T._records(); arms2 = T._second_arms(recs)
    os.environ["AF_EXPERIMENT"] = "1"
    rows = {a[0]: [] for a in ARITH}
    for seed in sorted(recs):
        rec = recs[seed]
        video = T._video(seed, rec)
        ref_mid = np.mean([rec["psnr_at"][5000]] + ([arms2[seed][1][5000]] if seed in arms2 else []))
        ref_end = np.mean([rec["psnr_end"]] + ([arms2[seed][2]] if seed in arms2 else []))
        for name, mlp, dw in ARITH:
            for k, v in (("AF_MLP_MODE", mlp), ("AF_DW_MODE", dw)):
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
            p_pre, p_at, p_end, _ = T._run(seed, rec, None, rec["iters"], video)
            rows[name].append((p_at[5000] - ref_mid, p_end - ref_end, p_at[5000], p_end))
            print("seed %d %-40s PSNR after 5000 iterations %.4f (reference %.4f), at the end %.4f (reference %.4f)" % (seed, name, p_at[5000], ref_mid, p_end, ref_end), flush=True)
    n = len(recs)
    print()
    for name, r in rows.items():
        r = np.array(r)
        print("%-40s hip - reference over %d seeds: after 5000 iterations %+.4f +- %.4f dB, at the end %+.4f +- %.4f dB"
              % (name, n, r[:, 0].mean(), r[:, 0].std(ddof=1) / np.sqrt(n), r[:, 1].mean(), r[:, 1].std(ddof=1) / np.sqrt(n)))
    base = np.array(rows[ARITH[2][0]])
    for name in (ARITH[0][0], ARITH[1][0]):
        d = np.array(rows[name])[:, 3] - base[:, 3]
        print("%-40s minus the fp32-MFMA arithmetic, paired over the seeds, at the end: %+.4f +- %.4f dB" % (name, d.mean(), d.std(ddof=1) / np.sqrt(n)))


if __name__ == "__main__":
    main()
