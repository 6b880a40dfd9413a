"""sks_forward_backward against sks_forward + sks_backward, interleaved on one box: the H36M API step (4 x 1000 x 1000), one rank's
share of the 8-GPU Panoptic step (4 x 1920 x 1080) and the whole 31-view Panoptic step.   python tools/ab_one_call.py [reps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from skelsplat_amd import rasterizer as R

dev = torch.device("cuda:0")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 7


def timed(fn, n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return 1e6 * (time.perf_counter() - t0) / n


for name, wl, ids, n in (("h36m 4 views", bench.WORKLOADS["h36m"], None, 300), ("panoptic rank 0 of 8 (4 views)", bench.WORKLOADS["panoptic"], [0, 8, 16, 24], 200),
                         ("panoptic 31 views", bench.WORKLOADS["panoptic"], None, 40)):
    if os.environ.get("ONLY") and os.environ["ONLY"] not in name:
        continue
    scene, gm, params = bench.make_scene(torch, wl, dev)
    cams = scene.cameras if ids is None else [scene.cameras[v] for v in ids]
    views = R.ViewBatch.from_cameras(cams)
    dL = torch.randn((len(cams), scene.n_joints, scene.H, scene.W), device=dev)
    steps = {k: bench.ApiStep(views, params, dL, one_call=k) for k in (False, True)}
    steps["graph"] = None
    for k in (False, True):
        for _ in range(5):
            steps[k]()
    try:
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            steps[True]()
        steps["graph"] = g.replay
    except Exception as e:
        print("graph capture failed:", repr(e)[:200])
        del steps["graph"]
    res = {k: [] for k in steps}
    for _ in range(reps):
        for k, fn in steps.items():
            res[k].append(timed(fn, n))
    for k, v in res.items():
        v.sort()
        print(f"{name}: {'one call' if k is True else 'two calls' if k is False else 'one call, hipGraph'}: median {v[len(v) // 2]:.1f} us  (min {v[0]:.1f}, max {v[-1]:.1f})", flush=True)
    del steps, dL
    torch.cuda.empty_cache()
