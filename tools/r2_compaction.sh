#!/bin/bash
# compaction of the flow-match rows: GPU suite, then points/s and per-kernel times at several valid fractions (single and two-layer)
set -u
OUT=gpurun_out; mkdir -p $OUT
timeout 1200 python -m pytest tests -m gpu -q --tb=short -x --deselect tests/test_gpu_c1.py 2>&1 | tail -6
for vf in 1.0 0.7 0.5; do
  timeout 300 python bench.py --no-cpu-baseline --valid-fraction $vf > $OUT/cmp_s_$vf.json 2> $OUT/cmp_s_$vf.err
  timeout 300 python bench.py --no-cpu-baseline --two-layer --valid-fraction $vf > $OUT/cmp_t_$vf.json 2> $OUT/cmp_t_$vf.err
done
python - <<PY
import json
for f in ["cmp_s_1.0","cmp_s_0.7","cmp_s_0.5","cmp_t_1.0","cmp_t_0.7","cmp_t_0.5"]:
    try:
        d=json.loads(open("$OUT/%s.json"%f).read().strip().splitlines()[-1]); r=d["roofline"]
        print(f, round(d["value"]), round(d["ms_per_step"],4), round(r["valid_flow_fraction"],3), {k: round(v["ms_per_step"],4) for k,v in r["by_kernel"].items()})
    except Exception as e:
        print(f, "FAILED", e, open("$OUT/%s.err"%f).read()[-800:])
PY
