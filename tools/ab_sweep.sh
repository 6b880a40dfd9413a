#!/bin/bash
# interleaved A/B of library builds on one box with tools/width_sweep2.py:  bash tools/ab_sweep.sh "<widths>" "<tunes>" lib1.so lib2.so ...
W=$1; T=$2; shift; shift
for rep in 1 2; do
  for L in "$@"; do
    echo "== $L"
    SKS_LIB_OVERRIDE=$PWD/$L python3 tools/width_sweep2.py $W $T 2>&1 | grep -v amdgpu.ids | grep rep0
  done
done
