#!/usr/bin/env python3
"""Host-side cost of one API step (sks_forward + sks_backward through rasterizer.py): cProfile over many steps, and the
step rate with the GPU work removed from the picture (tiny image: kernels of a few microseconds)."""
import cProfile, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from skelsplat_amd import rasterizer as R

dev = torch.device("cuda", 0)
wl = dict(dataset="h36m", V=4, name="x")
scene, gm, params = bench.make_scene(torch, wl, dev)
views = R.ViewBatch.from_cameras(scene.cameras)
dL = torch.randn((4, 17, scene.H, scene.W), device=dev)
for tag, step in (("workspace", bench.ApiStep(views, params, dL)),):
    for _ in range(50):
        step()
    torch.cuda.synchronize()
    n = 2000
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    t_issue = time.perf_counter() - t0          # host time to ISSUE n steps (the queue may run ahead of the GPU or not)
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print(f"{tag}: issue {1e6 * t_issue / n:.1f} us/step, complete {1e6 * t_all / n:.1f} us/step")
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(n):
        step()
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr)
    st.sort_stats("tottime").print_stats(18)
