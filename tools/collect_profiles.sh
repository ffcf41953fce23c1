#!/bin/bash
# Collect the rocprofv3 evidence for profiles/ on a GPU box (run through gpurun from the repo root):
#   kernel trace + stats of bench.py (single and two-layer), then one PMC pass each for FETCH_SIZE, WRITE_SIZE and
#   the SQ counters (separate passes: FETCH_SIZE takes 3 of the 4 TCC slots, MI355X_MICROARCH.md "rocprofv3 PMC slots").
# Usage: tools/collect_profiles.sh <tag>      -> gpurun_out/<tag>_*.txt
set -u
TAG=${1:-r2}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
# no warm-up, no pre-train: every k_dw / k_mlp_* dispatch in the trace belongs to bench.py's three passes over the same 40 iterations (20 with 9, 20 with
# 7 row segments): the timed region, the per-kernel pass, and the pass with the opt-in three-product weight-gradient GEMM (k_dw_bf<3>, + 5 warm-up steps)
BENCH="python $ROOT/bench.py --no-cpu-baseline --steps 40 --warmup 0 --pretrain-iters 0 --settle-steps 0"      # no settle steps either (round 5): they would be 200 more 9-segment iterations in the averages
run() {  # name, rocprof args..., -- cmd
  local name=$1; shift
  rm -rf /tmp/prof_$name
  (cd /tmp && rocprofv3 "$@" -d /tmp/prof_$name --output-format csv -- $CMD > $OUT/${TAG}_${name}.bench.json 2> $OUT/${TAG}_${name}.err)
  python $ROOT/tools/rocprof_summary.py /tmp/prof_$name > $OUT/${TAG}_${name}.txt
  local st=$(find /tmp/prof_$name -name "*kernel_stats.csv" | head -1)
  [ -n "$st" ] && cp $st $OUT/${TAG}_${name}_kernel_stats.csv
}
CMD="$BENCH"
run trace --kernel-trace --stats
run pmc_fetch --kernel-trace --pmc FETCH_SIZE
run pmc_write --kernel-trace --pmc WRITE_SIZE
run pmc_sq --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE
CMD="$BENCH --two-layer"
run two_layer_trace --kernel-trace --stats
CMD="$BENCH --two-layer"
run two_layer_pmc_fetch --kernel-trace --pmc FETCH_SIZE
# un-profiled reference line of the same command
$BENCH > $OUT/${TAG}_bench_unprofiled.json 2>/dev/null
python $ROOT/tools/traffic_from_pmc.py /tmp/prof_pmc_fetch /tmp/prof_pmc_write > $OUT/${TAG}_traffic.json
ls -la $OUT | tail -20
