#!/bin/bash
# Sweep the two constants of the k_dw split-K cost model (host.hip build_sched) on a GPU box: per-row-tile overhead of
# the 8x8 jobs (AF_DW_OVH) and of the small-shape jobs (AF_DW_OVH_S).  Prints points/s, k_dw ms, k_dw fraction of peak.
for ovh in ${OVH:-20 40 50 60 80}; do for ovhs in ${OVHS:-20 30 40}; do
AF_DW_OVH=$ovh AF_DW_OVH_S=$ovhs python bench.py --no-cpu-baseline --steps 60 --warmup 10 ${BENCH_ARGS:-} | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ovh $ovh small $ovhs', round(d['value']), round(d['roofline']['kernel_ms'],4), round(d['roofline']['frac'],4))"
done; done
