#!/bin/bash
# round-2 GPU call: bf16x6 chains — correctness vs the fp32 chains, timing of both, all new Z tests, full GPU suite
set -u
OUT=gpurun_out; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_mlp_modes.py tests/test_gpu_dw_modes.py -q -s --tb=short -x > $OUT/r2d_modes.log 2>&1; tail -3 $OUT/r2d_modes.log
timeout 300 python bench.py --no-cpu-baseline --steps 40 --warmup 10 > $OUT/r2d_bench.json 2> $OUT/r2d_bench.err
AF_MLP_FP32=1 timeout 300 python bench.py --no-cpu-baseline --steps 40 --warmup 10 > $OUT/r2d_bench_mlpfp32.json 2> $OUT/r2d_bench_mlpfp32.err
timeout 300 python bench.py --no-cpu-baseline --steps 40 --warmup 10 --two-layer > $OUT/r2d_bench_two_layer.json 2> $OUT/r2d_bench_two_layer.err
timeout 1500 python -m pytest tests -m gpu -q -s --tb=short --deselect tests/test_gpu_dw_modes.py --deselect tests/test_gpu_mlp_modes.py > $OUT/r2d_pytest.log 2>&1
tail -8 $OUT/r2d_pytest.log
for f in r2d_bench r2d_bench_mlpfp32 r2d_bench_two_layer; do python - <<PY
import json
try:
    d=json.load(open("$OUT/$f.json")); r=d["roofline"]
    print("$f", round(d["value"]), "pts/s", round(d["ms_per_step"],4), "ms/step", {k: round(v,4) for k,v in r["warmup_ms_per_step_by_kernel"].items()})
except Exception as e:
    print("$f", "FAILED", e); print(open("$OUT/$f.err").read()[-1500:])
PY
done
grep "forward max\|gradient rel\|loss terms" $OUT/r2d_modes.log | head -40
