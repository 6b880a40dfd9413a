#!/bin/bash
# Interleaved A/B of library builds on the frame-batched loop, one box:  bash tools/ab_frames.sh "<F list>" lib1.so lib2.so ...
Fs=$1; shift
for rep in 1 2 3; do
  for lib in "$@"; do
    echo "== $lib"
    SKS_LIB_OVERRIDE=$PWD/$lib ONLY_BATCH=1 python3 tools/bench_frames.py $Fs 2>&1 | grep "^F ="
  done
done
