#!/usr/bin/env python
"""Prints the rows of DESIGN.md section 4's table from the bench.py lines of a closing GPU call (tools/r4_final.sh):

    python tools/design_table.py <dir> <tag>                   bench lines <dir>/<tag>_bench*.json -> the throughput table
    python tools/design_table.py --pmc profiles/r4_pmc_sq.txt  every pipe-busy figure DESIGN.md quotes, from the committed SQ pass alone
    python tools/design_table.py --clock profiles/r4_step_clock.json   power / sclk while the loop runs and the in-kernel clocks of the step

(the text quoted figures the committed profile did not give.  No clock is derived from GRBM_GUI_ACTIVE any more:
GRBM / 8 / duration read 2.46 GHz for k_dw_bf<3>, above the part's nominal clock - it is not a clock.)"""
import json, sys, os, re


def pmc_table(path):
    """Per hot kernel of a `tools/rocprof_summary.py` SQ pass: duration, matrix-pipe busy share at the NOMINAL 2.4 GHz clock (1024 SIMDs),
    the same over GRBM_GUI_ACTIVE (128 SIMDs per XCD-level cycle), issue-active and instruction-wait shares of the wave cycles."""
    txt = open(path).read().splitlines()
    avg = {}
    for l in txt:
        m = re.match(r"^(.*?)\s+(\d+)\s+([0-9.]+)\s+([0-9.]+)\s+([0-9.]+)$", l)
        if m:
            avg[m.group(1).strip()] = float(m.group(3))
    cur, pm = None, {}
    for l in txt:
        if l.startswith("  ") and not l.startswith("      "):
            cur = l.strip(); pm[cur] = {}
        m = re.match(r"^\s{6}(\w+)\s+(\d+)\s+\(n=(\d+)\)", l)
        if m and cur:
            pm[cur][m.group(1)] = float(m.group(2))
    print("| kernel | avg us | MFMA busy / (1024 SIMDs x duration x 2.4 GHz) | MFMA busy / (GRBM_GUI_ACTIVE x 128) | SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES | SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES |")
    print("|---|---|---|---|---|---|")
    for k, c in pm.items():
        if not k.startswith(("void k_mlp", "k_mlp", "void k_dw", "k_dw")) or "SQ_VALU_MFMA_BUSY_CYCLES" not in c:
            continue
        us = next((v for n, v in avg.items() if n.startswith(k[:40])), None)
        if not us or not c.get("GRBM_GUI_ACTIVE"):
            continue
        print("| `%s` | %.1f | %.2f | %.2f | %.2f | %.2f |" % (k.replace("void ", ""), us, c["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * us * 2400.0),
              c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["GRBM_GUI_ACTIVE"] * 128.0), c.get("SQ_ACTIVE_INST_ANY", 0) / c["SQ_WAVE_CYCLES"], c.get("SQ_WAIT_INST_ANY", 0) / c["SQ_WAVE_CYCLES"]))


def clock_table(path):
    j = json.load(open(path))
    r = j["rocm_smi"]
    print("%s: %.4f ms/step; rocm-smi at %.1f Hz over the loop (%d samples): socket power median %.0f W (min %.0f, max %.0f), sclk median %.0f MHz (min %s, max %s); before the loop %s W / %s MHz"
          % (j["workload"], j["ms_per_step"], r["sample_rate_hz"], r["samples_while_running"], r["power_w"]["median"], r["power_w"]["min"], r["power_w"]["max"],
             r["sclk_mhz"]["median"], r["sclk_mhz"]["min"], r["sclk_mhz"]["max"], r["idle_before"]["power_w"], r["idle_before"]["sclk_mhz"]))
    print("| launch of the last step | workgroups (with work) | launch span us | workgroup span us (mean) | s_memtime ticks per workgroup | ticks / s_memrealtime span = clock MHz (mean, min .. max) |")
    print("|---|---|---|---|---|---|")
    for k, v in j["in_kernel_clock_of_the_last_step"].items():
        print("| %s | %d (%d) | %.1f | %.1f | %.0f | %.0f (%.0f .. %.0f) |" % (k, v["workgroups"], v["workgroups_with_work"], v["launch_span_us"], v["workgroup_span_us_mean"],
              v["ticks_per_workgroup_mean"], v["clock_mhz_mean"], v["clock_mhz_min"], v["clock_mhz_max"]))


if len(sys.argv) > 2 and sys.argv[1] == "--pmc":
    pmc_table(sys.argv[2]); sys.exit(0)
if len(sys.argv) > 2 and sys.argv[1] == "--clock":
    clock_table(sys.argv[2]); sys.exit(0)
OUT = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out"
TAG = sys.argv[2] if len(sys.argv) > 2 else "r4"
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
