#!/bin/bash
# Settle-the-clock harness: three readings of the same sustained back-to-back k_dw_bf loop:
#   (1) s_memtime ticks per workgroup / launch time (printed by dwbench built with -DDW_CLK),
#   (2) rocm-smi sclk / power sampled at ~5 Hz while the loop runs,
#   (3) GRBM_GUI_ACTIVE / duration comes from the rocprofv3 PMC pass (tools/collect_profiles.sh), not from here.
# Usage: tools/clock_sample.sh <binary> <args...>   -> stdout: the binary's output, then the samples
OUT=$(mktemp)
( while true; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|fclk|socclk|Power" | tr -s ' ' | tr '\n' ';'; echo; sleep 0.15; done ) > $OUT &
SP=$!
"$@"
kill $SP 2>/dev/null
wait $SP 2>/dev/null
echo "--- rocm-smi samples while the loop ran ---"
sort $OUT | uniq -c | sort -rn | head -12
rm -f $OUT
