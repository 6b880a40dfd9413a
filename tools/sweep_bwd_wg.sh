#!/bin/bash
# Workgroups per (view, Gaussian) of the wave-resident backward (SKS_BWD_WG = 1..5 -> 16, 8, 4, 2, 1) across frame-batch sizes.
for wg in 1 2 3 4; do
  echo "== SKS_BWD_WG=$wg  ($((16 >> (wg - 1))) workgroups per pair)"
  SKS_BWD_WG=$wg ONLY_BATCH=1 python3 tools/bench_frames.py ${1:-1 2 4 8 16} 2>&1 | grep "^F ="
done
