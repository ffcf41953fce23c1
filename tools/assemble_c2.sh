#!/bin/bash
# Build container, after the configs[1] reference runs of round 6 (.c2runs/, started by hand: seven single-thread runs of
# `oracle/make_golden_c1.py --resx 768 --resy 432 --iters 10001 --log-every 250 --psnr-at 5000 --seeds k --flow <kind> --threads 1`) have finished:
# first arms of seeds 5..8 join tests/golden/c2_reference.npz, second arms of seeds 0, 2, 4 join tests/golden/c2_reference_rerun.npz.
set -e
G=tests/golden; S=.c2runs
cp $G/c2_reference.npz /tmp/c2_ref_before.npz; cp $G/c2_reference_rerun.npz /tmp/c2_rerun_before.npz
PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden_c1.py --merge /tmp/c2_ref_before.npz $S/c2_s5_arm1_t1.npz $S/c2_s6_arm1_t1.npz $S/c2_s7_arm1_t1.npz $S/c2_s8_arm1_t1.npz --out $G/c2_reference.npz
PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden_c1.py --merge /tmp/c2_rerun_before.npz $S/c2_s0_arm2_t1.npz $S/c2_s2_arm2_t1.npz $S/c2_s4_arm2_t1.npz --out $G/c2_reference_rerun.npz
python - <<'PY'
import numpy as np
for f in ("c2_reference", "c2_reference_rerun"):
    d = np.load("tests/golden/%s.npz" % f)
    print(f, "seeds", d["seeds"], "threads", d["threads_per_seed"], "flow", d["flow_kind"], "PSNR end", np.round(d["psnr"], 3), "at 5000", np.round(d["psnr_at"][:, 0], 3), "cpu h", np.round(d["cpu_seconds"].sum(1) / 3600, 1))
PY
# ... and the second arm of seed 6 (eight threads), run after the first comparison showed that seed 0.5 dB apart at the switch:
#   python oracle/make_golden_c1.py --resx 768 --resy 432 --iters 10001 --log-every 250 --psnr-at 5000 --seeds 6 --flow field --threads 8 --out .c2runs/c2_s6_arm2_t8.npz
#   python oracle/make_golden_c1.py --merge tests/golden/c2_reference_rerun.npz .c2runs/c2_s6_arm2_t8.npz --out tests/golden/c2_reference_rerun.npz
