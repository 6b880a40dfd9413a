#!/bin/bash
# HBM traffic of the binned path's kernels (tools/bench_stress.py): PMC WRITE_SIZE and FETCH_SIZE in separate passes,
# with --kernel-trace only (MI355X_MICROARCH.md, HBM / rocprofv3 section).  bash tools/pmc_traffic_stress.sh   (GPU box)
root=$(cd "$(dirname "$0")/.." && pwd)
out=${1:-$root/gpurun_out/prof}
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$out/stress_w" -o w -- python3 "$root/tools/bench_stress.py" > /dev/null 2> "$out/stress_w.log"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$out/stress_f" -o f -- python3 "$root/tools/bench_stress.py" > /dev/null 2> "$out/stress_f.log"
python3 - "$out" <<'PY'
import csv, glob, collections, sys
for tag, counter, mul in (("w", "WRITE_SIZE", 1.0), ("f", "FETCH_SIZE", 2.0)):
    acc = collections.defaultdict(list)
    for f in glob.glob(sys.argv[1] + f"/stress_{tag}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                acc[r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:60]].append(float(r["Counter_Value"]))
    for k, v in sorted(acc.items()):
        if k.startswith("k_"):
            print(f"{counter:10s} {k:62s} n={len(v):4d}  {mul * 1024 * sum(v) / len(v) / 1e6:10.2f} MB/launch" + (" (x2 gfx950 correction)" if mul == 2 else ""))
PY
