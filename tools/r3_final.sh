#!/bin/bash
# round-3 closing GPU call: full GPU suite, smoke, the profiles/ evidence (tools/collect_profiles.sh), the bench lines and the complete schedules quoted in DESIGN.md §4
set -u
OUT=gpurun_out; mkdir -p $OUT
[ -z "${SKIP_PYTEST:-}" ] && timeout 2400 python -m pytest tests -m gpu -q -s --tb=short > $OUT/r3f_pytest.log 2>&1
tail -5 $OUT/r3f_pytest.log
grep -E "^seed|^mean PSNR|^reference against|hip - reference|device sampler|worst" $OUT/r3f_pytest.log | cut -c1-300 | head -60
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/r3f_smoke.log 2>&1; tail -2 $OUT/r3f_smoke.log
timeout 900 bash tools/collect_profiles.sh r3 > $OUT/r3f_collect.log 2>&1
timeout 400 python bench.py > $OUT/r3_bench.json 2> $OUT/r3_bench.err
timeout 300 python bench.py --no-cpu-baseline --two-layer > $OUT/r3_bench_two_layer.json 2> $OUT/r3_bench_two_layer.err
timeout 300 python bench.py --no-cpu-baseline --valid-fraction 0.7 > $OUT/r3_bench_valid07.json 2> $OUT/r3_bench_valid07.err
timeout 300 python bench.py --no-cpu-baseline --valid-fraction 0.5 > $OUT/r3_bench_valid05.json 2> $OUT/r3_bench_valid05.err
timeout 300 python bench.py --no-cpu-baseline --two-layer --valid-fraction 0.7 > $OUT/r3_bench_two_layer_valid07.json 2> $OUT/r3_bench_two_layer_valid07.err
timeout 300 python bench.py --no-cpu-baseline --steps 8000 --warmup 50 > $OUT/r3_bench_8000.json 2> $OUT/r3_bench_8000.err
python tools/show_bench.py $OUT/r3_bench.json $OUT/r3_bench_8000.json $OUT/r3_bench_two_layer.json $OUT/r3_bench_valid07.json $OUT/r3_bench_valid05.json $OUT/r3_bench_two_layer_valid07.json $OUT/r3_bench_unprofiled.json
timeout 300 python tools/full_run.py > $OUT/r3_full_run_single.json 2> $OUT/r3_full_run_single.err; tail -1 $OUT/r3_full_run_single.json | cut -c1-600
timeout 400 python tools/full_run.py --two-layer > $OUT/r3_full_run_two_layer.json 2> $OUT/r3_full_run_two_layer.err; tail -1 $OUT/r3_full_run_two_layer.json | cut -c1-600
timeout 900 python tools/full_run.py --frames 200 --resx 1920 --resy 1080 --iters 100000 > $OUT/r3_full_run_200f_1080p_100k.json 2> $OUT/r3_full_run_200f.err; tail -1 $OUT/r3_full_run_200f_1080p_100k.json | cut -c1-600
timeout 600 python tools/cli_end_to_end.py > $OUT/r3_cli_single.log 2>&1; tail -3 $OUT/r3_cli_single.log | cut -c1-400
timeout 600 python tools/cli_end_to_end.py --two-layer > $OUT/r3_cli_two_layer.log 2>&1; tail -3 $OUT/r3_cli_two_layer.log | cut -c1-400
