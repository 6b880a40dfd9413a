#!/bin/bash
# interleaved A/B of library builds on one box with tools/width_sweep2.py:  bash tools/ab_sweep.sh "<widths>" lib1.so lib2.so ...
W=$1; shift
for rep in 1 2; do
  for L in "$@"; do
    echo "== $L"
    SKS_LIB_OVERRIDE=$PWD/$L python3 tools/width_sweep2.py $W 0 2>&1 | grep -v amdgpu.ids | grep rep0
  done
done
