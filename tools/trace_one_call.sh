#!/bin/bash
# Kernel timeline of one API step, the two calls and the one call (rocprofv3 --kernel-trace).   bash tools/trace_one_call.sh [h36m|panoptic4]
root=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
for oc in 0 1; do
rm -rf /tmp/toc && WL=${1:-h36m} ONE_CALL=$oc rocprofv3 --kernel-trace --output-format csv -d /tmp/toc -o t -- python3 "$root/tools/one_call_step.py" > /dev/null 2>&1
echo "ONE_CALL=$oc"
python3 - <<'PY'
import csv, glob
rows = []
for f in glob.glob("/tmp/toc/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:44], r.get("Queue_Id", "?")))
rows.sort()
starts = [i for i, r in enumerate(rows) if r[2].startswith("k_geom_fwd")]
i0 = starts[-6]
t0 = rows[i0][0]
for r in rows[i0:i0 + 13]:
    print(f"{(r[0]-t0)/1e3:9.1f} -> {(r[1]-t0)/1e3:9.1f} us  ({(r[1]-r[0])/1e3:7.1f})  q{r[3]}  {r[2]}")
PY
done
