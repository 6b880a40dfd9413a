#!/bin/bash
# Stress backward: library variants (workgroups per CU in the launch bounds) x workgroups per CU in the launch (GPU box):
#   bash tools/sweep_bwd_tile.sh "ab_a.so ab_b.so" "32 64 128"
root=$(cd "$(dirname "$0")/.." && pwd)
for rep in 1 2; do
  for lib in $1; do
    for bpc in $2; do
      echo -n "$lib blocks/CU $bpc: "; SKS_BWD_TILE_BLOCKS=$((256 * bpc)) SKS_LIB_OVERRIDE=$root/skelsplat_amd/$lib python3 $root/tools/stress_kernels.py 1 2>/dev/null
    done
  done
done
