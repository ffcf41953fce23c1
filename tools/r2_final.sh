#!/bin/bash
# round-2 closing GPU call: full GPU suite, smoke, the profiles/ evidence (tools/collect_profiles.sh) and the bench lines quoted in DESIGN.md §4
set -u
OUT=gpurun_out; mkdir -p $OUT
timeout 1800 python -m pytest tests -m gpu -q -s --tb=short > $OUT/r2f_pytest.log 2>&1
tail -5 $OUT/r2f_pytest.log
grep -E "^seed|^mean PSNR|^reference against|total loss every" $OUT/r2f_pytest.log | head -30
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/r2f_smoke.log 2>&1; tail -2 $OUT/r2f_smoke.log
timeout 900 bash tools/collect_profiles.sh r2 > $OUT/r2f_collect.log 2>&1
timeout 400 python bench.py > $OUT/r2_bench.json 2> $OUT/r2_bench.err
timeout 300 python bench.py --no-cpu-baseline --two-layer > $OUT/r2_bench_two_layer.json 2> $OUT/r2_bench_two_layer.err
timeout 300 python bench.py --no-cpu-baseline --valid-fraction 0.7 > $OUT/r2_bench_valid07.json 2> $OUT/r2_bench_valid07.err
timeout 300 python bench.py --no-cpu-baseline --valid-fraction 0.5 > $OUT/r2_bench_valid05.json 2> $OUT/r2_bench_valid05.err
timeout 300 python bench.py --no-cpu-baseline --two-layer --valid-fraction 0.7 > $OUT/r2_bench_two_layer_valid07.json 2> $OUT/r2_bench_two_layer_valid07.err
timeout 300 python bench.py --no-cpu-baseline --steps 8000 --warmup 50 > $OUT/r2_bench_8000.json 2> $OUT/r2_bench_8000.err
for f in r2_bench r2_bench_8000 r2_bench_two_layer r2_bench_valid07 r2_bench_valid05 r2_bench_two_layer_valid07 r2_bench_unprofiled; do python - <<PY
import json
try:
    d=json.loads(open("$OUT/$f.json").read().strip().splitlines()[-1]); r=d["roofline"]
    print("$f", round(d["value"]), "pts/s", round(d["ms_per_step"],4), "ms/step", r["kernel"], r["bound"], round(r["frac"],3), {k: round(v["ms_per_step"],4) for k,v in r["by_kernel"].items()})
except Exception as e:
    print("$f", "FAILED", e)
PY
done
