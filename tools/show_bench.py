#!/usr/bin/env python
"""Print the headline and the per-kernel table of one or more bench.py JSON lines."""
import json, sys
for f in sys.argv[1:]:
    d = json.load(open(f))
    print("%s: %.3f M points/s, %.4f ms/step, bf16x3-dW %s ms ; dominant %s frac %.3f" % (f, d["value"] / 1e6, d["ms_per_step"], d.get("ms_per_step_bf16x3_dw"), d["roofline"]["kernel"], d["roofline"]["frac"]))
    for k, v in d["roofline"]["by_kernel"].items():
        if isinstance(v, dict):
            print("    %-28s %.4f ms/step" % (k, v["ms_per_step"]))
