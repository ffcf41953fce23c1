#!/bin/bash
# round-4 closing GPU call: full GPU suite, smoke, the profiles/ evidence (tools/collect_profiles.sh), the bench lines and the complete schedules quoted in DESIGN.md §4
set -u
OUT=gpurun_out; mkdir -p $OUT
[ -z "${SKIP_PYTEST:-}" ] && timeout 2400 python -m pytest tests -m gpu -q -s --tb=short > $OUT/r4f_pytest.log 2>&1
tail -5 $OUT/r4f_pytest.log
grep -E "^seed|^mean PSNR|^reference against|hip - reference|device sampler|worst" $OUT/r4f_pytest.log | cut -c1-300 | head -60
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/r4f_smoke.log 2>&1; tail -2 $OUT/r4f_smoke.log
timeout 900 bash tools/collect_profiles.sh r4 > $OUT/r4f_collect.log 2>&1
timeout 400 python bench.py > $OUT/r4_bench.json 2> $OUT/r4_bench.err
timeout 300 python bench.py --no-cpu-baseline --two-layer > $OUT/r4_bench_two_layer.json 2> $OUT/r4_bench_two_layer.err
timeout 300 python bench.py --no-cpu-baseline --valid-fraction 0.7 > $OUT/r4_bench_valid07.json 2> $OUT/r4_bench_valid07.err
timeout 300 python bench.py --no-cpu-baseline --valid-fraction 0.5 > $OUT/r4_bench_valid05.json 2> $OUT/r4_bench_valid05.err
timeout 300 python bench.py --no-cpu-baseline --two-layer --valid-fraction 0.7 > $OUT/r4_bench_two_layer_valid07.json 2> $OUT/r4_bench_two_layer_valid07.err
timeout 300 python bench.py --no-cpu-baseline --steps 8000 --warmup 50 > $OUT/r4_bench_8000.json 2> $OUT/r4_bench_8000.err
python tools/show_bench.py $OUT/r4_bench.json $OUT/r4_bench_8000.json $OUT/r4_bench_two_layer.json $OUT/r4_bench_valid07.json $OUT/r4_bench_valid05.json $OUT/r4_bench_two_layer_valid07.json $OUT/r4_bench_unprofiled.json
timeout 300 python tools/full_run.py > $OUT/r4_full_run_single.json 2> $OUT/r4_full_run_single.err; tail -1 $OUT/r4_full_run_single.json | cut -c1-600
timeout 400 python tools/full_run.py --two-layer > $OUT/r4_full_run_two_layer.json 2> $OUT/r4_full_run_two_layer.err; tail -1 $OUT/r4_full_run_two_layer.json | cut -c1-600
timeout 900 python tools/full_run.py --frames 200 --resx 1920 --resy 1080 --iters 100000 > $OUT/r4_full_run_200f_1080p_100k.json 2> $OUT/r4_full_run_200f.err; tail -1 $OUT/r4_full_run_200f_1080p_100k.json | cut -c1-600
timeout 600 python tools/cli_end_to_end.py > $OUT/r4_cli_single.log 2>&1; tail -3 $OUT/r4_cli_single.log | cut -c1-400
timeout 600 python tools/cli_end_to_end.py --two-layer > $OUT/r4_cli_two_layer.log 2>&1; tail -3 $OUT/r4_cli_two_layer.log | cut -c1-400
# the clock and the power inside the real step: rocm-smi at >= 10 Hz across 8000 steps + per-workgroup s_memtime / s_memrealtime of the five hot launches
timeout 300 python tools/step_clock.py --steps 8000 > $OUT/r4_step_clock.json 2> $OUT/r4_step_clock.err; python tools/design_table.py --clock $OUT/r4_step_clock.json
timeout 300 python tools/step_clock.py --steps 4000 --two-layer > $OUT/r4_step_clock_two_layer.json 2>> $OUT/r4_step_clock.err; python tools/design_table.py --clock $OUT/r4_step_clock_two_layer.json
# k_dw in isolation: the compiler-scheduled 8x8 stage against the slotted one (same results bit for bit: the hash), ticks per stage and clock of a sustained loop
for b in dwb_slot0 dwb_slot1; do [ -x tools/bin/$b ] && { echo "== $b"; timeout 60 tools/bin/$b 66 0 10; timeout 60 tools/bin/$b 66 1 10 | grep -v "mode 0\|hash [0-9a-f]*$"; }; done > $OUT/r4_dwbench.txt 2>&1
python tools/design_table.py --pmc $OUT/r4_pmc_sq.txt
