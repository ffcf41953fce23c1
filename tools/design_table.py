#!/usr/bin/env python
"""Prints the rows of DESIGN.md section 4's table from the bench.py lines of a closing GPU call (tools/r3_final.sh)."""
import json, sys, os
OUT = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out"
TAG = sys.argv[2] if len(sys.argv) > 2 else "r3"
rows = [("configs[1] single atlas, iterations 4901..5100", "bench"), ("… iterations 1001..9000 (8 000 timed steps)", "bench_8000"),
        ("configs[4] two-layer (`--two-layer`)", "bench_two_layer"), ("configs[1], `--valid-fraction 0.7`", "bench_valid07"),
        ("configs[1], `--valid-fraction 0.5`", "bench_valid05"), ("configs[4], `--valid-fraction 0.7`", "bench_two_layer_valid07")]
for label, name in rows:
    p = os.path.join(OUT, "%s_%s.json" % (TAG, name))
    try:
        d = json.loads(open(p).read().strip().splitlines()[-1])
    except Exception as e:
        print("| %s | (missing: %s) |" % (label, e)); continue
    bk = d["roofline"]["by_kernel"]
    g = lambda pre: next((v["ms_per_step"] for k, v in bk.items() if isinstance(v, dict) and k.startswith(pre)), float("nan"))
    dw = next((v for k, v in bk.items() if isinstance(v, dict) and k.startswith("k_dw")), {})
    print("| %s (`profiles/%s_%s.json`) | **%.2f M** | %.3f | %.3f | %.3f | %.3f ms (%.2f TB/s) | %.2f M / %.3f ms |"
          % (label, TAG, name, d["value"] / 1e6, d["ms_per_step"], g("k_mlp_fwd"), g("k_mlp_bwd"), g("k_dw"), dw.get("algorithmic_hbm_gbs", 0) / 1e3,
             d.get("value_bf16x3_dw", 0) / 1e6, d.get("ms_per_step_bf16x3_dw", 0)))
    if name == "bench" and "cpu_baseline" in d:
        print("| host CPU, oracle port, %s threads | %.1f k | | | | | |" % (d["cpu_baseline"].get("cores"), (d["cpu_baseline"].get("value") or 0) / 1e3))
