#!/bin/bash
# Round 5, GPU call 1: the new parity tests, the energy budget, the CLI wall clock, the partition-sigma A/B, a bench line.
cd "${GRAFT_REPO_ROOT:-.}"; O=gpurun_out/r5a; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_seg.py -x -q -m gpu -s -k "render or step_clocks or timing or error" > $O/pytest_render.log 2>&1; echo "render rc=$?" | tee -a $O/rc.txt
timeout 600 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -s -k "strict" > $O/pytest_strict.log 2>&1; echo "strict rc=$?" | tee -a $O/rc.txt
timeout 600 python -m pytest tests/test_stage1_host.py -x -q -m gpu > $O/pytest_host.log 2>&1; echo "host rc=$?" | tee -a $O/rc.txt
timeout 600 python tools/energy_budget.py > $O/energy.json 2> $O/energy.err; echo "energy rc=$?" | tee -a $O/rc.txt
timeout 300 python tools/cli_end_to_end.py > $O/cli_single.json 2> $O/cli_single.err; echo "cli rc=$?" | tee -a $O/rc.txt
timeout 300 python tools/cli_end_to_end.py --two-layer > $O/cli_two_layer.json 2> $O/cli_two_layer.err; echo "cli2 rc=$?" | tee -a $O/rc.txt
timeout 300 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" | tee -a $O/rc.txt
timeout 900 python tools/partition_sigma.py > $O/partition_sigma.json 2> $O/partition_sigma.err; echo "sigma rc=$?" | tee -a $O/rc.txt
tail -3 $O/pytest_render.log $O/pytest_strict.log $O/pytest_host.log; cat $O/cli_single.json $O/cli_two_layer.json; tail -c 600 $O/bench.json
