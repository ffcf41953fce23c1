"""Summarise a rocprofv3 --kernel-trace [--pmc ...] --output-format csv run into a small text table
(average duration per kernel, call counts, PMC averages) for committing under profiles/."""
import collections
import csv
import glob
import re
import sys

d = sys.argv[1]
out = []
kt = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
if kt:
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(kt[0])):
        agg[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    tot = sum(sum(v) for v in agg.values())
    out.append("%-60s %8s %12s %12s %7s" % ("kernel", "calls", "avg_us", "total_us", "%"))
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        if sum(v) / tot < 0.002:
            continue
        out.append("%-60s %8d %12.2f %12.1f %7.2f" % (k[:60], len(v), sum(v) / len(v), sum(v), 100 * sum(v) / tot))
cc = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
if cc:
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(cc[0])):
        agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out.append("")
    out.append("PMC averages per dispatch (kernels of this repo only)")
    for k, cs in agg.items():
        if not re.match(r"^(void )?k_", k):       # this repository's kernels (templated ones print as "void k_...<...>(...)")
            continue
        out.append("  " + k[:70])
        for c, v in sorted(cs.items()):
            out.append("      %-32s %16.0f   (n=%d)" % (c, sum(v) / len(v), len(v)))
print("\n".join(out))
