#!/bin/bash
# Round 6, GPU call 2: where the ticks of an f16x3 chain go (hfbench), the chaos spread of the two 10-iteration trajectory tests over
# arithmetics x split-K partitions, the torch third opinion on the input builder, the two-rank rehearsal on one device.
cd "${GRAFT_REPO_ROOT:-.}"; O=gpurun_out/r6f; mkdir -p $O
export TMPDIR=/tmp
tools/bin/hfbench > $O/hfbench.txt 2>&1; echo "hfbench rc=$?" | tee -a $O/rc.txt
timeout 300 python -m pytest tests/test_gpu_builder_torch.py -q -m gpu -s > $O/pytest_builder_torch.log 2>&1; echo "builder rc=$?" | tee -a $O/rc.txt
for mode in 1 3 0; do for cost in "" "306,150,126,129,87" "306,170,145,148,100"; do
  echo "=== AF_MLP_MODE=$mode AF_DW_COST=$cost" >> $O/chaos_seg_small.log
  AF_EXPERIMENT=1 AF_MLP_MODE=$mode AF_DW_COST=$cost timeout 300 python -m pytest "tests/test_gpu_seg.py::test_trajectory_psnr_and_parameters_match_reference" -q -m gpu -s 2>&1 | grep -E "max rel|passed|failed" >> $O/chaos_seg_small.log
  echo "=== AF_MLP_MODE=$mode AF_DW_COST=$cost" >> $O/chaos_seg_full.log
  AF_EXPERIMENT=1 AF_MLP_MODE=$mode AF_DW_COST=$cost timeout 600 python -m pytest "tests/test_gpu_fullsize.py::test_full_size_seg_trajectory_matches_oracle" -q -m gpu -s 2>&1 | grep -E "max rel|passed|failed" >> $O/chaos_seg_full.log
done; done
AF_BENCH_SHARE_DEVICE=1 timeout 600 python bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_two_ranks_one_device.json 2> $O/bench_two_ranks_one_device.err; echo "two ranks rc=$?" | tee -a $O/rc.txt
cat $O/hfbench.txt; tail -3 $O/pytest_builder_torch.log; grep -c passed $O/chaos_seg_small.log $O/chaos_seg_full.log; tail -c 800 $O/bench_two_ranks_one_device.json; tail -5 $O/bench_two_ranks_one_device.err
