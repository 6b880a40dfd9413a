"""The H36M API step through sks_forward_backward alone (for rocprofv3):  ONE_CALL=0 for the two separate calls, WL=panoptic4 for one
rank's share of the 8-GPU Panoptic step."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from skelsplat_amd import rasterizer as R

dev = torch.device("cuda:0")
wl = os.environ.get("WL", "h36m")
scene, gm, params = bench.make_scene(torch, bench.WORKLOADS["panoptic" if wl == "panoptic4" else wl], dev)
cams = [scene.cameras[v] for v in (0, 8, 16, 24)] if wl == "panoptic4" else scene.cameras
views = R.ViewBatch.from_cameras(cams)
dL = torch.randn((len(cams), scene.n_joints, scene.H, scene.W), device=dev)
step = bench.ApiStep(views, params, dL, one_call=os.environ.get("ONE_CALL", "1") != "0")
for _ in range(int(os.environ.get("STEPS", "40"))):
    step()
torch.cuda.synchronize()
