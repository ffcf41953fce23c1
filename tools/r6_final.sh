#!/bin/bash
# round-6 closing GPU calls.  Part "profiles": rocprofv3 evidence (tools/collect_profiles.sh), bench lines, complete schedules, in-step clock.
# Part "suite": the full GPU suite + smoke on the final tree (incl. the complete configs[1] schedules of tests/test_gpu_c2.py).
set -u
OUT=gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
if [ "${1:-all}" != "suite" ]; then
  timeout 900 bash tools/collect_profiles.sh r6 > $OUT/r6f_collect.log 2>&1
  timeout 400 python bench.py > $OUT/r6_bench.json 2> $OUT/r6_bench.err
  timeout 300 python bench.py --no-cpu-baseline --two-layer > $OUT/r6_bench_two_layer.json 2> $OUT/r6_bench_two_layer.err
  timeout 300 python bench.py --no-cpu-baseline --steps 8000 --warmup 50 > $OUT/r6_bench_8000.json 2> $OUT/r6_bench_8000.err
  timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 > $OUT/r6_bench_driver_style.json 2> $OUT/r6_bench_driver_style.err
  python tools/show_bench.py $OUT/r6_bench.json $OUT/r6_bench_8000.json $OUT/r6_bench_two_layer.json $OUT/r6_bench_driver_style.json $OUT/r6_bench_unprofiled.json
  timeout 300 python tools/full_run.py > $OUT/r6_full_run_single.json 2> $OUT/r6_full_run_single.err; tail -1 $OUT/r6_full_run_single.json | cut -c1-600
  timeout 400 python tools/full_run.py --two-layer > $OUT/r6_full_run_two_layer.json 2> $OUT/r6_full_run_two_layer.err; tail -1 $OUT/r6_full_run_two_layer.json | cut -c1-600
  timeout 300 python tools/step_clock.py --steps 8000 > $OUT/r6_step_clock.json 2> $OUT/r6_step_clock.err; python tools/design_table.py --clock $OUT/r6_step_clock.json
  python tools/design_table.py --pmc $OUT/r6_pmc_sq.txt
fi
if [ "${1:-all}" != "profiles" ]; then
  timeout 2400 python -m pytest tests -m gpu -q -s --tb=short --durations=12 > $OUT/r6f_pytest.log 2>&1
  tail -18 $OUT/r6f_pytest.log
  grep -E "^seed|^mean PSNR|^reference against|hip - reference|device sampler|worst|sigma of one run|PSNR after" $OUT/r6f_pytest.log | cut -c1-300 | head -80
  timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/r6f_smoke.log 2>&1; tail -2 $OUT/r6f_smoke.log
fi
