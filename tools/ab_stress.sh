#!/bin/bash
# Interleaved A/B of library variants on the stress config (one box):  bash tools/ab_stress.sh ab_x.so ab_y.so ...
root=$(cd "$(dirname "$0")/.." && pwd)
for rep in 1 2 3; do
  for lib in "$@"; do
    echo -n "$lib: "; SKS_LIB_OVERRIDE=$root/skelsplat_amd/$lib python3 $root/tools/stress_kernels.py 1 2>/dev/null
  done
done
