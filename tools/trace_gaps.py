"""Idle time between consecutive kernels of a rocprofv3 --kernel-trace CSV (one stream): per-step sum and distribution."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
rows = sorted(r for r in rows if r[2].startswith("k_") or "k_mlp" in r[2] or "k_adam" in r[2])
gaps = {}
steps = 0
for a, b in zip(rows, rows[1:]):
    g = (b[0] - a[1]) / 1e3
    if g > 200: continue                      # host-side pauses (setup, sync)
    key = a[2].split("(")[0][:22] + " -> " + b[2].split("(")[0][:22]
    gaps.setdefault(key, []).append(g)
    steps += a[2].startswith("k_dw")
tot = 0
for k, v in sorted(gaps.items(), key=lambda kv: -sum(kv[1])):
    print("%-50s n=%4d  mean %6.2f us  max %6.2f" % (k, len(v), sum(v) / len(v), max(v))); tot += sum(v)
print("steps %d, idle between kernels per step: %.1f us" % (steps, tot / max(steps, 1)))
