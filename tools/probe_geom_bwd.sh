#!/bin/bash
root=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
for v in plain raw raw_idle plain_idle raw_feat v4_raw raw_sparse; do
  rm -rf /tmp/pg && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pg -o t -- python3 "$root/tools/probe_geom_bwd.py" $v > /tmp/pg.log 2>&1
  tail -1 /tmp/pg.log
  python3 "$root/tools/kstats.py" /tmp/pg 8 | grep "k_geom_bwd\|k_render_bwd"
done
