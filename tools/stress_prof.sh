#!/bin/bash
# Per-kernel averages + one forward/backward timeline of the binned stress config (GPU box):  bash tools/stress_prof.sh [outfile]
root=$(cd "$(dirname "$0")/.." && pwd)
out=${1:-$root/gpurun_out/stress_prof.txt}
case "$out" in /*) ;; *) out="$PWD/$out" ;; esac
mkdir -p "$(dirname "$out")"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/sp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sp -o t -- python3 "$root/tools/bench_stress.py" > /tmp/sp.log 2>&1
{
tail -4 /tmp/sp.log
python3 "$root/tools/kstats.py" /tmp/sp 14
python3 - <<'PY'
import csv, glob
rows = []
for f in glob.glob("/tmp/sp/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:44]))
rows.sort()
starts = [i for i, r in enumerate(rows) if r[2].startswith("k_geom_fwd")]
i0 = starts[-3]      # inside the fwd+bwd loop
t0 = rows[i0][0]
for r in rows[i0 - 1:i0 + 12]:
    print(f"{(r[0]-t0)/1e3:9.1f} -> {(r[1]-t0)/1e3:9.1f} us  ({(r[1]-r[0])/1e3:7.1f})  {r[2]}")
PY
} > "$out" 2>&1
cat "$out"
