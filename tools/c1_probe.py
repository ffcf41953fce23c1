#!/usr/bin/env python
"""Final PSNR of one seed of the configs[0] / configs[4] complete schedule through the HIP path (the GPU tests' own runner), for probing how
far a different summation order (another split-K partition: AF_DW_COST, another library build: AF_LIB_PATH) moves the END of the chaotic
trajectory.  Usage: python tools/c1_probe.py [--seg] seed [seed ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
seg = "--seg" in sys.argv
seeds = [int(a) for a in sys.argv[1:] if a.lstrip("-").isdigit()]
if seg:
    import test_gpu_c1_seg as T
    g = dict(np.load(T.GOLDEN))
else:
    import test_gpu_c1 as T
    g = dict(np.load(T.GOLDEN)); m = dict(np.load(T.MORE))
for s in seeds:
    src = g if (seg or s in [int(x) for x in g["seeds"]]) else m
    r = T._run(s, src, True)
    print("seed %d %s: PSNR after pre-train %.4f, final %.4f  [AF_DW_COST=%s AF_LIB_PATH=%s]" % (s, "seg" if seg else "single", r[0], r[1], os.environ.get("AF_DW_COST"), os.path.basename(os.environ.get("AF_LIB_PATH", "default"))), flush=True)
