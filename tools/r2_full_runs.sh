#!/bin/bash
# the complete schedules, end to end (tools/full_run.py) and the drop-in CLI from files on disk (tools/cli_end_to_end.py)
set -u
OUT=gpurun_out; mkdir -p $OUT
timeout 300 python tools/full_run.py > $OUT/r2_full_run_single.json 2> $OUT/r2_full_run_single.err; tail -1 $OUT/r2_full_run_single.json | cut -c1-600
timeout 400 python tools/full_run.py --two-layer > $OUT/r2_full_run_two_layer.json 2> $OUT/r2_full_run_two_layer.err; tail -1 $OUT/r2_full_run_two_layer.json | cut -c1-600
timeout 900 python tools/full_run.py --frames 200 --resx 1920 --resy 1080 --iters 100000 > $OUT/r2_full_run_200f_1080p_100k.json 2> $OUT/r2_full_run_200f.err; tail -1 $OUT/r2_full_run_200f_1080p_100k.json | cut -c1-600
timeout 600 python tools/cli_end_to_end.py > $OUT/r2_cli_single.log 2>&1; tail -3 $OUT/r2_cli_single.log | cut -c1-400
timeout 600 python tools/cli_end_to_end.py --two-layer > $OUT/r2_cli_two_layer.log 2>&1; tail -3 $OUT/r2_cli_two_layer.log | cut -c1-400
