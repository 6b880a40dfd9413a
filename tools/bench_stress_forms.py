#!/usr/bin/env python3
"""Stress scene (BASELINE configs[4]): the step as two calls and as one call (sks_forward_backward: view groups pipelined over two
streams, SKS_BIN_GROUPS groups -- the library reads the variable once per process), interleaved on one box.
    SKS_BIN_GROUPS=4 python tools/bench_stress_forms.py [reps] [aux priority: 0 | -1]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from skelsplat_amd import rasterizer as R
from skelsplat_amd.scene import stress_scene

dev = torch.device("cuda", 0)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
V = 8
sc, g = stress_scene(V)
views = R.ViewBatch.from_cameras([cam.to(dev) for cam in sc.cameras])
args = (t(g["means"]), t(g["feat"]), t(g["opac"]), t(g["scales"]), t(g["quats"]), None)
dL = torch.randn((V, 17, 2048, 2048), device=dev)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
prio = int(sys.argv[2]) if len(sys.argv) > 2 else 0
ws2, ws1 = R.Workspace(), R.Workspace()
if prio:
    ws1._aux[dev.index] = torch.cuda.Stream(device=dev.index, priority=prio)


def two():
    st = R.forward_views(views, *args, bin_capacity=400000, workspace=ws2, check_capacity="auto")[3]
    return R.backward_views(st, *args, dL, workspace=ws2)["means3D"]


def one():
    return R.forward_backward_views(views, *args, dL, workspace=ws1, bin_capacity=400000, check_capacity="auto")[4]["means3D"]


for fn in (two, one):
    for _ in range(4):
        fn()
torch.cuda.synchronize()
a_, b_ = two().clone(), one().clone()
torch.cuda.synchronize()
print("one call == two calls bit for bit:", bool(torch.equal(a_, b_)), " max |diff| / max |grad|:",
      float((a_ - b_).abs().max() / a_.abs().max()))
res = {"two": [], "one": []}
n = 20
for _ in range(reps):
    for tag, fn in (("two", two), ("one", one)):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        res[tag].append(1e3 * (time.perf_counter() - t0) / n)
med = {k: sorted(v)[len(v) // 2] for k, v in res.items()}
print(f"groups={os.environ.get('SKS_BIN_GROUPS', 'default')} aux_priority={prio}: two calls {med['two']:.4f} ms, one call {med['one']:.4f} ms "
      f"(ratio {med['one'] / med['two']:.3f}, fused backward {os.environ.get('SKS_FUSED_BWD', '1')}); reps two {[round(x, 4) for x in res['two']]} one {[round(x, 4) for x in res['one']]}")
