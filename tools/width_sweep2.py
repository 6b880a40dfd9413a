#!/usr/bin/env python3
"""Forward kernel time across widths, with the skeleton in view and with every Gaussian culled (pure fill: no band has a
rect, no composite block has work): what the covered bands cost."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from skelsplat_amd import rasterizer as R, _lib
from skelsplat_amd.scene import SyntheticScene, GaussianModel

dev = torch.device("cuda", 0)
widths = [int(w) for w in (sys.argv[1].split(",") if len(sys.argv) > 1 else "1000,1002,1024".split(","))]
tunes = [int(x, 0) for x in (sys.argv[2].split(",") if len(sys.argv) > 2 else ["0"])]
H = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
for rep in range(2):
    for W in widths:
        sc = SyntheticScene("h36m", n_views=4, seed=0, device=dev, W=W, H=H, fx=1145.0)
        gm = GaussianModel().create_from_points(sc.pose_3d_init, sc.spatial_lr_scale, 17, device=dev)
        with torch.no_grad():
            args = [gm.get_xyz.detach().clone(), gm.get_features.reshape(17, 17).contiguous(), gm.get_opacity.detach(),
                    gm.get_scaling.detach(), gm.get_rotation.detach(), None]
        views = R.ViewBatch.from_cameras(sc.cameras)
        for culled in (False, True):
            a = list(args)
            if culled:
                a[0] = a[0] + 1e9
            for tune in tunes:
                ws = R.Workspace()
                for _ in range(10):
                    R.forward_views(views, *a, workspace=ws, tune_flags=tune)
                torch.cuda.synchronize()
                _lib.prof_enable(True, every=1)
                _lib.prof_read(0)
                for _ in range(60):
                    R.forward_views(views, *a, workspace=ws, tune_flags=tune)
                torch.cuda.synchronize()
                ms, n, q = _lib.prof_read_quantiles(0)
                _lib.prof_enable(False)
                print(f"W={W:5d} culled={int(culled)} tune={tune:#x} rep{rep}: {1e3*ms/n:6.2f} us  p50 {1e3*q[1]:6.2f}", flush=True)
                del ws
