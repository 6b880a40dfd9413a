"""Why is k_geom_bwd 29 us in the drop-in iteration (rocprofv3, tools/dropin_trace.sh) and 5-6 us everywhere else?
Runs the raw backward for one H36M view under a few switches; rocprofv3 --kernel-trace --stats gives the kernel's duration
per variant (each variant is a separate process argument: python3 tools/probe_geom_bwd.py <variant>)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from skelsplat_amd import _lib
from skelsplat_amd import rasterizer as R
from skelsplat_amd.scene import SyntheticScene, GaussianModel

variant = sys.argv[1] if len(sys.argv) > 1 else "plain"
dev = torch.device("cuda:0")
V = 4 if "v4" in variant else 1
scene = SyntheticScene("h36m", n_views=4, seed=0, device=dev)
gm = GaussianModel().create_from_points(scene.pose_3d_init, scene.spatial_lr_scale, scene.n_joints, scene_type="h36m", device=dev)
views = R.ViewBatch.from_cameras(scene.cameras[:V])
raw = "raw" in variant
if raw:
    args = (gm._xyz.detach(), gm.get_features.detach(), gm._opacity.detach(), gm._scaling.detach(), gm._rotation.detach(), None)
    flags = _lib.SKS_RAW_PARAMS | _lib.SKS_RAW_GRADS
else:
    args = (gm.get_xyz.detach(), gm.get_features.detach(), gm.get_opacity.detach(), gm.get_scaling.detach(), gm.get_rotation.detach(), None)
    flags = 0
col, inv, radii, st = R.forward_views(views, *args, clamp01=True, tune_flags=flags)
dL = torch.randn_like(col)
if "sparse" in variant:
    dL = dL * (col > 0)
idle = "idle" in variant
for i in range(60):
    g = R.backward_views(st, *args, dL, want_dfeatures="feat" in variant, tune_flags=flags)
    if idle:
        torch.cuda.synchronize()
        time.sleep(0.0005)
torch.cuda.synchronize()
print(variant, "done", float(g["means3D"].abs().sum()))
