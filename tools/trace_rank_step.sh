#!/bin/bash
# Timeline of one rank step (tools/bench_rank_step.py under rocprofv3 --kernel-trace).   bash tools/trace_rank_step.sh   (GPU box)
root=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/trr && STEPS=40 rocprofv3 --kernel-trace --output-format csv -d /tmp/trr -o t -- python3 "$root/tools/bench_rank_step.py" > /dev/null 2>&1
python3 - <<'PY'
import csv, glob
rows = []
for f in glob.glob("/tmp/trr/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:50], r.get("Queue_Id", "?")))
rows.sort()
starts = [i for i, r in enumerate(rows) if r[2].startswith("k_geom_fwd")]
i0 = starts[25]
t0 = rows[i0][0]
for r in rows[i0:i0 + 18]:
    print(f"{(r[0]-t0)/1e3:9.1f} -> {(r[1]-t0)/1e3:9.1f} us  ({(r[1]-r[0])/1e3:7.1f})  q{r[3]}  {r[2]}")
PY
