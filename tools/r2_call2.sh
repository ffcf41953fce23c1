#!/bin/bash
# round-2 GPU call 2: bf16x6 k_dw — correctness vs the fp32-MFMA k_dw, timing of both, full GPU suite
set -u
OUT=gpurun_out; mkdir -p $OUT
python -m pytest tests/test_gpu_dw_modes.py -q -s --tb=short > $OUT/r2b_dwmodes.log 2>&1; tail -3 $OUT/r2b_dwmodes.log
python bench.py --no-cpu-baseline --steps 40 --warmup 10 > $OUT/r2b_bench.json 2> $OUT/r2b_bench.err
AF_DW_FP32=1 python bench.py --no-cpu-baseline --steps 40 --warmup 10 > $OUT/r2b_bench_dwfp32.json 2> $OUT/r2b_bench_dwfp32.err
python bench.py --no-cpu-baseline --steps 40 --warmup 10 --two-layer > $OUT/r2b_bench_two_layer.json 2> $OUT/r2b_bench_two_layer.err
python -m pytest tests -m gpu -q -s --tb=short --deselect tests/test_gpu_dw_modes.py > $OUT/r2b_pytest.log 2>&1
tail -5 $OUT/r2b_pytest.log
for f in r2b_bench r2b_bench_dwfp32 r2b_bench_two_layer; do python - <<PY
import json
d=json.load(open("$OUT/$f.json")); r=d["roofline"]
print("$f", round(d["value"]), "pts/s", round(d["ms_per_step"],4), "ms/step", {k: round(v,4) for k,v in r["warmup_ms_per_step_by_kernel"].items()})
PY
done
grep "gradient rel" $OUT/r2b_dwmodes.log | head -20
