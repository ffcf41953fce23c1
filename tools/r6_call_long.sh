#!/bin/bash
# Round 6: configs[2] (200 frames 1920x1080, 100 000 iterations) complete schedule + its rocprofv3 trace / PMC passes, CLI end to end (both paths)
cd "${GRAFT_REPO_ROOT:-.}"; O=gpurun_out; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python tools/full_run.py --frames 200 --resx 1920 --resy 1080 --iters 100000 > $O/r6_full_run_200f_1080p_100k.json 2> $O/r6_full_run_200f.err; tail -1 $O/r6_full_run_200f_1080p_100k.json | cut -c1-600
B="python $PWD/bench.py --no-cpu-baseline --steps 40 --warmup 0 --pretrain-iters 0 --settle-steps 0 --frames 200 --resx 1920 --resy 1080"
for pass in trace pmc_fetch pmc_write; do
  rm -rf /tmp/prof_$pass
  case $pass in trace) A="--kernel-trace --stats";; pmc_fetch) A="--kernel-trace --pmc FETCH_SIZE";; pmc_write) A="--kernel-trace --pmc WRITE_SIZE";; esac
  (cd /tmp && timeout 600 rocprofv3 $A -d /tmp/prof_$pass --output-format csv -- $B > $OLDPWD/$O/r6_200f_1080p_${pass}.bench.json 2> $OLDPWD/$O/r6_200f_1080p_${pass}.err)
  python tools/rocprof_summary.py /tmp/prof_$pass > $O/r6_200f_1080p_${pass}.txt 2>&1
done
python tools/traffic_from_pmc.py /tmp/prof_pmc_fetch /tmp/prof_pmc_write > $O/r6_200f_1080p_hbm_bytes_per_launch.json 2>&1
timeout 300 $B > $O/r6_200f_1080p_bench_unprofiled.json 2>/dev/null
timeout 300 python tools/cli_end_to_end.py > $O/r6_cli_single.json 2> $O/r6_cli_single.err; tail -1 $O/r6_cli_single.json | cut -c1-400
timeout 300 python tools/cli_end_to_end.py --two-layer > $O/r6_cli_two_layer.json 2> $O/r6_cli_two_layer.err; tail -1 $O/r6_cli_two_layer.json | cut -c1-400
head -8 $O/r6_200f_1080p_trace.txt; python tools/show_bench.py $O/r6_200f_1080p_bench_unprofiled.json
