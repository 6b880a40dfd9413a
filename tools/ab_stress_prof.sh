#!/bin/bash
# Per-kernel rocprofv3 averages of the stress step for library variants (GPU box):  bash tools/ab_stress_prof.sh ab_x.so ab_y.so ...
root=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  rm -rf /tmp/sp_$lib && SKS_LIB_OVERRIDE=$root/skelsplat_amd/$lib rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sp_$lib -o t -- python3 "$root/tools/bench_stress.py" > /tmp/sp.log 2>&1
  echo "== $lib"; python3 "$root/tools/kstats.py" /tmp/sp_$lib 12 | grep "k_"
done
