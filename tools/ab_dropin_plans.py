#!/usr/bin/env python3
"""The literal drop-in iteration (render -> fused criterion -> loss.backward() -> skelsplat_amd.optim.Adam every 4th view) with and
without the autograd path's recorded C-ABI calls:  for P in 0 1; do SKS_AUTOGRAD_PLANS=$P python tools/ab_dropin_plans.py; done
(run the pair several times in turn on one box: hosts change gear)."""
import os, sys, time, types
sys.path.insert(0, os.getcwd())
import torch
from gaussian_renderer import render_functions
from skelsplat_amd.heatmaps import generate_heatmaps
from skelsplat_amd.ops import l2_loss_gaussian
from skelsplat_amd.scene import SyntheticScene, GaussianModel
from skelsplat_amd.optim import Adam
dev = torch.device("cuda:0")
scene = SyntheticScene("h36m", n_views=4, seed=0, device=dev)
gm = GaussianModel().create_from_points(scene.pose_3d_init, scene.spatial_lr_scale, scene.n_joints, scene_type="h36m", device=dev)
gm.training_setup()
groups = [{k: g[k] for k in ("params", "lr", "name")} for g in gm.optimizer.param_groups]
gm.optimizer = Adam(groups, lr=0.0, eps=1e-15)
hm = generate_heatmaps(gm._xyz.detach(), gm.get_scaling.detach(), gm._rotation.detach(), torch.tensor(scene.poses_2d, device=dev), scene.cameras)
render = render_functions["diff-gaussian-rasterization-h36m"]
pipe = types.SimpleNamespace(debug=False, antialiasing=False, compute_cov3D_python=False, convert_SHs_python=False)
bg = torch.zeros(3, device=dev)
def it(i):
    pkg = render(scene.cameras[i % 4], gm, pipe, bg)
    loss, _ = l2_loss_gaussian(pkg["render"], hm[i % 4])
    loss.backward()
    if (i + 1) % 4 == 0:
        gm.optimizer.step(); gm.optimizer.zero_grad(set_to_none=True)
for i in range(64): it(i)
reps = []
for _ in range(5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(256): it(i)
    torch.cuda.synchronize(); reps.append(1e3 * (time.perf_counter() - t0) / 256)
print("SKS_AUTOGRAD_PLANS=" + os.environ.get("SKS_AUTOGRAD_PLANS", "1"), "drop-in iteration (fused criterion, one-launch Adam) ms per view:", sorted(round(r, 4) for r in reps))
