#!/bin/bash
# Build container, after the reference-module runs of tools/cpu_queue.py have written their per-seed files to $1 (default /tmp/seeds):
# stack them into the committed fixtures (tests/golden/) - the configs[0] single-arm seeds, the fp64 / grad64 arms, the configs[4] runs.
S=${1:-/tmp/seeds}
G=tests/golden
set -e
# only the seeds the committed fixture does not hold yet (the script may run again as further runs finish)
c1=$(python - $S <<'PY'
import glob, sys, numpy as np
have = {int(s) for s in np.load("tests/golden/c1_reference_more.npz")["seeds"]}
print(" ".join(f for f in sorted(glob.glob(sys.argv[1] + "/c1_s*.npz")) if not ({int(s) for s in np.load(f)["seeds"]} & have)))
PY
)
if [ -n "$c1" ]; then
  cp $G/c1_reference_more.npz /tmp/c1_more_before.npz
  PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden_c1.py --merge /tmp/c1_more_before.npz $c1 --out $G/c1_reference_more.npz
fi
for f in $S/c1_fp64_s*.npz; do [ -e "$f" ] && cp $f $G/c1_reference_fp64_seed$(basename $f .npz | sed 's/c1_fp64_s//').npz; done
for f in $S/c1_grad64_s*.npz; do [ -e "$f" ] && cp $f $G/c1_reference_grad64_seed$(basename $f .npz | sed 's/c1_grad64_s//').npz; done
seg=$(ls $S/c1seg_s*.npz 2>/dev/null | tr '\n' ' ')
if [ -n "$seg" ]; then
  PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden_c1_seg.py --merge gpurun_out/c1seg/s*_t*.npz $seg --out $G/c1_seg_reference.npz
fi
ls -la $G | grep c1_
