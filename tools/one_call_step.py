"""One workload's API step alone (for rocprofv3; bench.py's measure_traffic runs it under --pmc passes): WL = h36m | panoptic |
panoptic4 (one rank's share of the 8-GPU Panoptic step) | stress (BASELINE configs[4], binned path); ONE_CALL=1 for the step through
sks_forward_backward instead of the two separate calls; STEPS."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from skelsplat_amd import rasterizer as R

dev = torch.device("cuda:0")
wl = os.environ.get("WL", "h36m")
one_call = os.environ.get("ONE_CALL", "0") == "1"
steps = int(os.environ.get("STEPS", "40"))
if wl == "stress":
    import numpy as np
    from skelsplat_amd.scene import stress_scene
    sc, g = stress_scene(8)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    views = R.ViewBatch.from_cameras([cam.to(dev) for cam in sc.cameras])
    params = (t(g["means"]), t(g["feat"]), t(g["opac"]), t(g["scales"]), t(g["quats"]))
    dL = torch.randn((8, 17, 2048, 2048), device=dev)
    ws = R.Workspace()
    for _ in range(steps):
        if one_call:
            R.forward_backward_views(views, *params, None, dL, workspace=ws, bin_capacity=400000)
        else:
            st = R.forward_views(views, *params, None, bin_capacity=400000, workspace=ws)[3]
            R.backward_views(st, *params, None, dL, workspace=ws)
else:
    scene, gm, params = bench.make_scene(torch, bench.WORKLOADS["panoptic" if wl == "panoptic4" else wl], dev)
    cams = [scene.cameras[v] for v in (0, 8, 16, 24)] if wl == "panoptic4" else scene.cameras
    views = R.ViewBatch.from_cameras(cams)
    dL = torch.randn((len(cams), scene.n_joints, scene.H, scene.W), device=dev)
    step = bench.ApiStep(views, params, dL, one_call=one_call)
    for _ in range(steps):
        step()
torch.cuda.synchronize()
