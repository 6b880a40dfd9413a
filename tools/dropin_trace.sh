#!/bin/bash
# Per-kernel averages + the timeline of one literal drop-in iteration (render -> criterion -> backward) under rocprofv3 (GPU box):
#   bash tools/dropin_trace.sh [--tensor] [outfile]
root=$(cd "$(dirname "$0")/.." && pwd)
flag=""
if [ "$1" = "--tensor" ]; then flag="--tensor"; shift; fi
out=${1:-$root/gpurun_out/dropin_trace.txt}
case "$out" in /*) ;; *) out="$PWD/$out" ;; esac
mkdir -p "$(dirname "$out")"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/dt && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/dt -o t -- python3 "$root/tools/profile_dropin.py" $flag --no-torch-profiler > /tmp/dt.log 2>&1
{
grep "sync=" /tmp/dt.log
python3 "$root/tools/kstats.py" /tmp/dt 24
python3 - <<'PY'
import csv, glob
rows = []
for f in glob.glob("/tmp/dt/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:70]))
rows.sort()
starts = [i for i, r in enumerate(rows) if r[2].startswith("k_geom_fwd")]
i0, i1 = starts[20], starts[21]      # one iteration inside the unsynchronised loop
t0 = rows[i0][0]
for r in rows[i0:i1 + 1]:
    print(f"{(r[0]-t0)/1e3:9.1f} -> {(r[1]-t0)/1e3:9.1f} us  ({(r[1]-r[0])/1e3:7.1f})  {r[2]}")
PY
} > "$out" 2>&1
cat "$out"
