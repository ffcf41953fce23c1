#!/usr/bin/env python
"""Where does this path's run-to-run sigma on configs[4] come from?

The complete fg/bg schedule (80 x 160x90, 2 x 8000 pre-train steps, 1001 iterations, the reference run's video / weights / draws:
tests/test_gpu_c1_seg.py's runner) is repeated per seed over several split-K partitions of k_dw (af_debug_set_dw_cost: another
summation ORDER of the same partial products) in two arithmetics: the shipped bf16x6 chains + bf16x6 k_dw, and the fp32-MFMA twins
(mlp_mode 0, dw_mode 0).  If the pooled sigma over partitions is the same in both, it is the summation granularity of the split-K
reduction (any fp32 implementation with this partitioning would show it); if the fp32 twins scatter less, it is bf16x6.

    python tools/partition_sigma.py [--seeds 0 1 2] [--single]      -> one JSON object
"""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

PARTS = (None, "306,150,126,129,87", "306,170,145,148,100", "306,158,133,136,92", "306,166,140,143,97,90", "306,180,150,150,105")

ap = argparse.ArgumentParser()
ap.add_argument("--seeds", type=int, nargs="+", default=[0, 1, 2])
a = ap.parse_args()
import test_gpu_c1_seg as T
g = dict(np.load(T.GOLDEN))
out = {"partitions": [p or "shipped" for p in PARTS], "seeds": a.seeds, "psnr": {}, "reference_psnr": {}}
for s in a.seeds:
    out["reference_psnr"][s] = [float(p) for p, ss in zip(g["psnr"], g["seeds"]) if int(ss) == s]
for name, env in (("bf16x6", {}), ("fp32_mfma", {"AF_MLP_FP32": "1", "AF_DW_FP32": "1"})):
    for k in ("AF_MLP_FP32", "AF_DW_FP32"):
        os.environ.pop(k, None)
    os.environ.update(env)
    rows = []
    for s in a.seeds:
        rows.append([float(T._run(s, g, True, partition=p)[1]) for p in PARTS])
        print(name, "seed", s, np.array2string(np.array(rows[-1]), precision=3), file=sys.stderr, flush=True)
    r = np.array(rows)
    out["psnr"][name] = r.tolist()
    out.setdefault("sigma_pooled_db", {})[name] = float(np.sqrt(np.mean(r.var(axis=1, ddof=1))))
    out.setdefault("sigma_per_seed_db", {})[name] = r.std(axis=1, ddof=1).tolist()
for k in ("AF_MLP_FP32", "AF_DW_FP32"):
    os.environ.pop(k, None)
print(json.dumps(out))
