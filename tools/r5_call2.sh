#!/bin/bash
# Round 5, GPU call 2: render tests (fp64 yardstick), narrower nets, the unfenced build on the architectures that broke in round 3,
# the energy budget with a validated power source, more seeds of the partition-sigma A/B, rocprofv3 of configs[2] (200 x 1920x1080).
cd "${GRAFT_REPO_ROOT:-.}"; O=gpurun_out/r5b; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_seg.py -x -q -m gpu -s -k "render" > $O/pytest_render.log 2>&1; echo "render rc=$?" | tee -a $O/rc.txt
timeout 900 python -m pytest tests/test_gpu_arch.py -x -q -m gpu -s -k "narrower or widths" > $O/pytest_width.log 2>&1; echo "width rc=$?" | tee -a $O/rc.txt
AF_LIB_PATH=$PWD/tools/bin/libatlasfit_nofence.so timeout 900 python -m pytest tests/test_gpu_arch.py tests/test_gpu_fullsize.py -q -m gpu -k "non_shipped or long_run or every_layer_count" > $O/pytest_nofence.log 2>&1; echo "nofence rc=$?" | tee -a $O/rc.txt
timeout 600 python tools/energy_budget.py > $O/energy.json 2> $O/energy.err; echo "energy rc=$?" | tee -a $O/rc.txt
timeout 900 python tools/partition_sigma.py --seeds 3 4 5 6 7 > $O/partition_sigma_seeds3to7.json 2> $O/partition_sigma.err; echo "sigma rc=$?" | tee -a $O/rc.txt
# configs[2]: 200 frames 1920x1080 (26.5 GB record table), the bf16x6 kernels
B="python $PWD/bench.py --no-cpu-baseline --steps 40 --warmup 0 --pretrain-iters 0 --frames 200 --resx 1920 --resy 1080"
for pass in trace pmc_fetch pmc_write; do
  rm -rf /tmp/prof_$pass
  case $pass in trace) A="--kernel-trace --stats";; pmc_fetch) A="--kernel-trace --pmc FETCH_SIZE";; pmc_write) A="--kernel-trace --pmc WRITE_SIZE";; esac
  (cd /tmp && timeout 600 rocprofv3 $A -d /tmp/prof_$pass --output-format csv -- $B > $OLDPWD/$O/c2_${pass}.bench.json 2> $OLDPWD/$O/c2_${pass}.err)
  python tools/rocprof_summary.py /tmp/prof_$pass > $O/c2_200f_1080p_${pass}.txt 2>&1
done
python tools/traffic_from_pmc.py /tmp/prof_pmc_fetch /tmp/prof_pmc_write > $O/c2_200f_1080p_traffic.json 2>&1
timeout 300 $B > $O/c2_200f_1080p_bench_unprofiled.json 2>/dev/null
timeout 300 python bench.py --no-cpu-baseline --steps 40 --warmup 0 --pretrain-iters 0 > $O/c1_same_flags_bench.json 2>/dev/null
cat $O/rc.txt; tail -3 $O/pytest_render.log $O/pytest_width.log $O/pytest_nofence.log; head -12 $O/c2_200f_1080p_trace.txt
