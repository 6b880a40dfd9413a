#!/usr/bin/env python3
"""Kernel-level timing of the stress config without a profiler: hipEvent pairs of the library's own hook (sks_prof_*) around the
forward / backward compositors + wall clock of the calls.  python tools/stress_kernels.py [reps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch, ctypes
from skelsplat_amd import rasterizer as R, _lib
from skelsplat_amd.scene import stress_scene
dev = torch.device("cuda", 0)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev) if a is not None else None
V = 8
sc, g = stress_scene(V)
views = R.ViewBatch.from_cameras([cam.to(dev) for cam in sc.cameras])
args = (t(g["means"]), t(g["feat"]), t(g["opac"]), t(g["scales"]), t(g["quats"]), None)
dL = torch.randn((V, 17, 2048, 2048), device=dev)
ws = R.Workspace()
lib = _lib.load()
def step():
    st = R.forward_views(views, *args, bin_capacity=400000, workspace=ws, check_capacity=False)[3]
    R.backward_views(st, *args, dL, workspace=ws)
for _ in range(5): step()
torch.cuda.synchronize()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
for r in range(reps):
    lib.sks_prof_enable(1)
    t0 = time.perf_counter()
    for _ in range(20): step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    out = []
    for kind in (0, 1):
        ms, n = ctypes.c_double(), ctypes.c_longlong()
        lib.sks_prof_read(kind, ctypes.byref(ms), ctypes.byref(n))
        out.append(1e3 * ms.value / max(n.value, 1))
    lib.sks_prof_enable(0)
    print(f"step {dt*1e3:.3f} ms  fwd kernel {out[0]:.1f} us  bwd kernel {out[1]:.1f} us", flush=True)
