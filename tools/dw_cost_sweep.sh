#!/bin/bash
# Sweep of k_dw's per-shape tile costs (AF_DW_COST, host.hip build_sched) on the real step: bench.py per candidate row, k_dw ms/step and step time.
#   tools/dw_cost_sweep.sh "306,133,113,108,72" "306,153,125,130,84" ...
for C in "$@"; do
  AF_DW_COST=$C python bench.py --steps 1500 --warmup 50 --no-cpu-baseline > /tmp/sw.json 2>/dev/null
  python - "$C" <<'P'
import json, sys
j = json.load(open("/tmp/sw.json")); bk = j["roofline"]["by_kernel"]
print("AF_DW_COST=%-28s %.4f ms/step  k_dw %.4f  fwd %.4f  bwd %.4f" % (sys.argv[1], j["ms_per_step"], bk["k_dw_bf<6>"]["ms_per_step"], bk["k_mlp_fwd_multi_bf<true>"]["ms_per_step"], bk["k_mlp_bwd_multi_bf"]["ms_per_step"]))
P
done
