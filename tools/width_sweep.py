#!/usr/bin/env python3
"""Forward kernel time of the H36M-shaped 4-view launch across image widths (same skeleton, same cameras scaled):
separates what the 16-byte half-masked fill of W % 4 == 2 costs from what a width does by itself."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from skelsplat_amd import rasterizer as R, _lib
from skelsplat_amd.scene import SyntheticScene, GaussianModel

dev = torch.device("cuda", 0)
widths = [int(w) for w in (sys.argv[1].split(",") if len(sys.argv) > 1 else "992,996,1000,1002,1004,1006,1008,1024".split(","))]
H = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
res = {}
for rep in range(2):
    for W in widths:
        sc = SyntheticScene("h36m", n_views=4, seed=0, device=dev, W=W, H=H, fx=1145.0)
        gm = GaussianModel().create_from_points(sc.pose_3d_init, sc.spatial_lr_scale, 17, device=dev)
        with torch.no_grad():
            args = (gm.get_xyz.detach(), gm.get_features.reshape(17, 17).contiguous(), gm.get_opacity.detach(),
                    gm.get_scaling.detach(), gm.get_rotation.detach(), None)
        views = R.ViewBatch.from_cameras(sc.cameras)
        ws = R.Workspace()
        for _ in range(10):
            R.forward_views(views, *args, workspace=ws)
        torch.cuda.synchronize()
        _lib.prof_enable(True, every=1)
        _lib.prof_read(0)
        for _ in range(60):
            R.forward_views(views, *args, workspace=ws)
        torch.cuda.synchronize()
        ms, n, q = _lib.prof_read_quantiles(0)
        _lib.prof_enable(False)
        us = 1e3 * ms / n
        res.setdefault(W, []).append(us)
        alg = 4.0 * H * W * 18 * 4
        print(f"W={W:5d} rep{rep}: {us:6.2f} us  p50 {1e3*q[1]:6.2f}  {alg/us/1e3:7.1f} GB/s", flush=True)
        del ws
