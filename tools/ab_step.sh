#!/bin/bash
# A/B of two builds of libatlasfit.so on the REAL step (bench.py), alternating, with rocm-smi power / clock samples at ~7 Hz while each runs:
#   tools/ab_step.sh <libA.so> <libB.so> [steps]
# Prints per run: points/s, ms/step, the per-kernel ms of by_kernel, and the most frequent rocm-smi readings.
A=$1; B=$2; K=${3:-3000}
for rep in 1 2; do
  for L in $A $B; do
    OUT=$(mktemp)
    ( while true; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr -s ' ' | tr '\n' ';'; echo; sleep 0.1; done ) > $OUT &
    SP=$!
    AF_LIB_PATH=$L python bench.py --steps $K --warmup 50 --no-cpu-baseline > /tmp/ab.json 2>/dev/null
    kill $SP 2>/dev/null; wait $SP 2>/dev/null
    echo "== $(basename $L) (rep $rep)"; python tools/show_bench.py /tmp/ab.json
    grep -o "Power[^;]*" $OUT | grep -o "[0-9.]* *W\|[0-9.]*$" | sort -n | awk '{a[NR]=$1} END {if (NR) printf("   power samples: n=%d median %s max %s\n", NR, a[int((NR+1)/2)], a[NR])}'
    grep -o "sclk[^;]*" $OUT | sort | uniq -c | sort -rn | head -3
    rm -f $OUT
  done
done
