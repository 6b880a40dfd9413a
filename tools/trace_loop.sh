#!/bin/bash
# Kernel timeline of the sparse training loop's accumulation groups (MultiViewLoop.step_group, H36M 4 views) under rocprofv3
# --kernel-trace: what one "grad step" (bench.py grad_step_ms) is made of.   [UNCHAINED=1] bash tools/trace_loop.sh   (GPU box)
root=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
cat > /tmp/trl.py <<PY
import sys; sys.path.insert(0, "$root")
import torch, bench
from skelsplat_amd.loop import MultiViewLoop
from skelsplat_amd.heatmaps import generate_heatmaps
dev = torch.device("cuda", 0)
wl = bench.WORKLOADS["h36m"]
scene, gm, params = bench.make_scene(torch, wl, dev)
gm.training_setup()
hm = generate_heatmaps(gm._xyz.detach(), gm.get_scaling.detach(), gm._rotation.detach(), torch.tensor(scene.poses_2d, device=dev), scene.cameras)
loop = MultiViewLoop(bench.fresh_model(scene, "h36m", dev), scene.cameras, hm, dataset="h36m", accumulation_steps=4, sparse=True)
if "$UNCHAINED" == "1":
    for _ in range(40):
        loop.step_group()          # each on its own: geometry from the parameters in front of every group
else:
    loop.run(160)                  # 40 groups, chained on the geometry the previous group's tail left
torch.cuda.synchronize()
PY
rm -rf /tmp/trl && rocprofv3 --kernel-trace --output-format csv -d /tmp/trl -o t -- python3 /tmp/trl.py > /dev/null 2>&1
python3 - <<'PY'
import csv, glob
rows = []
for f in glob.glob("/tmp/trl/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:50]))
rows.sort()
t0 = rows[-12][0]
for r in rows[-12:]:
    print(f"{(r[0]-t0)/1e3:9.1f} -> {(r[1]-t0)/1e3:9.1f} us  ({(r[1]-r[0])/1e3:7.1f})  {r[2]}")
PY
