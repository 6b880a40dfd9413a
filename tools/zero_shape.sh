#!/bin/bash
root=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/zs && rocprofv3 --kernel-trace --output-format csv -d /tmp/zs -o t -- python3 "$root/tools/zero_shape.py" > /dev/null 2>&1
python3 - <<'PY'
import csv, glob
for f in glob.glob("/tmp/zs/**/*kernel_trace.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    print(list(rows[0].keys()))
    for r in rows[-3:]:
        print({k: r[k] for k in r if k in ("Kernel_Name", "Workgroup_Size_X", "Grid_Size_X", "Grid_Size", "Workgroup_Size", "Start_Timestamp", "End_Timestamp", "LDS_Block_Size", "VGPR_Count", "SGPR_Count", "Accum_VGPR_Count", "Scratch_Size")}, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, "us")
PY
