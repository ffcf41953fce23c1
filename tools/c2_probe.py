#!/usr/bin/env python
"""Final PSNR of one seed of the complete configs[1] schedule through the HIP path (tests/test_gpu_c2.py's runner) under another arithmetic
(AF_MLP_FP32=1 AF_DW_FP32=1: the fp32-MFMA twins; AF_DW_MODE=2) or split-K partition (AF_DW_COST), to see whether a distance from the reference's
run follows the arithmetic.  Usage: [AF_MLP_FP32=1 AF_DW_FP32=1] python tools/c2_probe.py seed [seed ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_c2 as T
recs = T._records()
for s in [int(a) for a in sys.argv[1:]]:
    rec = recs[s]
    v = T._video(s, rec)
    p_pre, p_at, p_end, _ = T._run(s, rec, None, rec["iters"], v)
    print("seed %d (%s): PSNR after the pre-train %.4f (reference %.4f), at %s %s (reference %s), at the end %.4f (reference %.4f)   [AF_MLP_FP32=%s AF_DW_FP32=%s AF_DW_MODE=%s]"
          % (s, rec["flow"], p_pre, rec["psnr_pre"], list(p_at), [round(x, 4) for x in p_at.values()], [round(x, 4) for x in rec["psnr_at"].values()], p_end, rec["psnr_end"],
             os.environ.get("AF_MLP_FP32"), os.environ.get("AF_DW_FP32"), os.environ.get("AF_DW_MODE")), flush=True)
