#!/bin/bash
# round-2 GPU call 1: kernel ablations (old vs AGPR-fragment / interleaved schedule), MFMA ceiling, bench, parity tests
set -u
OUT=gpurun_out; mkdir -p $OUT
{
  tools/bin/abl_new_0 1024 1
  for b in old_0 new_0 afrag_0 sgb_0 old_1 new_1 old_2 new_2 old_3 new_3 old_15 new_15; do
    tools/bin/abl_$b 1024 2>&1 | grep map32
  done
  for b in old_0 new_0; do tools/bin/abl_$b 2813 2>&1 | grep map32; tools/bin/abl_$b 2188 2>&1 | grep map32; done
} > $OUT/r2a_ablate.txt 2>&1
python bench.py --no-cpu-baseline --steps 40 --warmup 10 > $OUT/r2a_bench.json 2> $OUT/r2a_bench.err
python bench.py --no-cpu-baseline --steps 40 --warmup 10 --two-layer > $OUT/r2a_bench_two_layer.json 2> $OUT/r2a_bench_two_layer.err
python -m pytest tests -m gpu -q -s --tb=short > $OUT/r2a_pytest.log 2>&1
tail -5 $OUT/r2a_pytest.log
cat $OUT/r2a_ablate.txt
