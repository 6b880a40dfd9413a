#!/bin/bash
# Binning-kernel durations of the stress scene for several builds of the library:  bash tools/trace_variants.sh libA.so libB.so ...
# (variants from tools/build_variant.sh; timing experiments only -- a variant may compute nonsense)
root=$(cd "$(dirname "$0")/.." && pwd)
for lib in "$@"; do
  export SKS_LIB_OVERRIDE=$root/$lib
  echo "== $lib"
  bash "$root/tools/trace_stress.sh" 2>&1 | grep -E "k_geom_fwd|k_bin|k_render_fwd" | tail -4
done
