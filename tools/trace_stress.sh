#!/bin/bash
# Timeline of one binned forward call (tools/bench_stress.py under rocprofv3 --kernel-trace): start / end of every kernel
# relative to the call's first kernel.   [TRACE_SCRIPT=bin_only.py] bash tools/trace_stress.sh   (GPU box)
root=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/trs && rocprofv3 --kernel-trace --output-format csv -d /tmp/trs -o t -- python3 "$root/tools/${TRACE_SCRIPT:-bench_stress.py}" > /dev/null 2>&1
python3 - <<'PY'
import csv, glob
rows = []
for f in glob.glob("/tmp/trs/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:44], r.get("Queue_Id", "?")))
rows.sort()
# the 12th forward call: find k_geom_fwd occurrences
starts = [i for i, r in enumerate(rows) if r[2].startswith("k_geom_fwd")]
i0 = starts[-6]
t0 = rows[i0][0]
for r in rows[i0:i0 + 12]:
    print(f"{(r[0]-t0)/1e3:9.1f} -> {(r[1]-t0)/1e3:9.1f} us  ({(r[1]-r[0])/1e3:7.1f})  q{r[3]}  {r[2]}")
PY
