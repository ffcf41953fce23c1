#!/bin/bash
# Round 6, GPU call 3: f16x3 as the library default with the element-wise tails finished inside the next block — tick breakdown, the mode / error /
# trajectory tests, the bench line.
cd "${GRAFT_REPO_ROOT:-.}"; O=gpurun_out/r6g; mkdir -p $O
export TMPDIR=/tmp
tools/bin/hfbench > $O/hfbench.txt 2>&1; echo "hfbench rc=$?" | tee -a $O/rc.txt
timeout 900 python -m pytest tests/test_gpu_mlp_modes.py tests/test_gpu_gemm_error.py tests/test_gpu_builder_torch.py -x -q -m gpu -s > $O/pytest_modes.log 2>&1; echo "modes rc=$?" | tee -a $O/rc.txt
timeout 900 python -m pytest tests/test_gpu_seg.py tests/test_gpu_parity.py tests/test_gpu_arch.py -q -m gpu -s > $O/pytest_small.log 2>&1; echo "small rc=$?" | tee -a $O/rc.txt
timeout 1200 python -m pytest tests/test_gpu_fullsize.py -q -m gpu -s > $O/pytest_fullsize.log 2>&1; echo "fullsize rc=$?" | tee -a $O/rc.txt
timeout 300 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" | tee -a $O/rc.txt
cat $O/hfbench.txt | grep -v ticks; tail -3 $O/pytest_modes.log $O/pytest_small.log $O/pytest_fullsize.log; python tools/show_bench.py $O/bench.json; tail -3 $O/bench.err
