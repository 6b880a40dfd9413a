#!/bin/bash
# rocprofv3 per-kernel averages of bench.py's API step for the H36M and Panoptic workloads (GPU box):  bash tools/kernel_stats_bench.sh
root=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
for WL in h36m panoptic; do
  rm -rf /tmp/kp_$WL; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kp_$WL -o s -- python3 "$root/bench.py" --workload $WL --steps 50 --warmup 5 --no-cpu-baseline --no-extras > /dev/null 2>&1
  echo "== $WL"; python3 "$root/tools/kstats.py" /tmp/kp_$WL 6 | grep "k_"
done
