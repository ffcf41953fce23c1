#!/bin/bash
# Builds the energy-budget harness variants (tools/energy.hip) into tools/bin/ (git-ignored; they travel to the GPU box with the snapshot).
cd "$(dirname "$0")/.." && mkdir -p tools/bin
F="--offload-arch=gfx950 -O3 -std=c++17"
for n in 0 1 2; do hipcc $F -DEN_DW -DDW_ABL=$n tools/energy.hip -o tools/bin/en_d$n || exit 1; done
for n in 0 1 2 8 32 43; do hipcc $F -DEN_CHAINS -DAF_ABL=$n tools/energy.hip -o tools/bin/en_c$n || exit 1; done
ls -la tools/bin/en_*
