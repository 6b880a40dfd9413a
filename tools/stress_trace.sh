#!/bin/bash
# timeline of one fwd+bwd of the stress config with a given library:  SKS_LIB_OVERRIDE=... bash tools/stress_trace.sh
root=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/st && rocprofv3 --kernel-trace --output-format csv -d /tmp/st -o t -- python3 "$root/tools/stress_kernels.py" 1 > /dev/null 2>&1
python3 - <<'PY'
import csv, glob
rows = []
for f in glob.glob("/tmp/st/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:44]))
rows.sort()
starts = [i for i, r in enumerate(rows) if r[2].startswith("k_geom_fwd")]
i0 = starts[-3]
t0 = rows[i0][0]
for r in rows[i0:i0 + 9]:
    print(f"{(r[0]-t0)/1e3:9.1f} -> {(r[1]-t0)/1e3:9.1f} us  ({(r[1]-r[0])/1e3:7.1f})  {r[2]}")
PY
