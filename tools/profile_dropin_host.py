"""cProfile of the literal drop-in iteration's HOST side (render -> fused criterion -> backward -> optimiser every 4th view)."""
import cProfile
import os
import pstats
import sys
import types

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gaussian_renderer import render_functions
from skelsplat_amd.heatmaps import generate_heatmaps
from skelsplat_amd.ops import l2_loss_gaussian
from skelsplat_amd.scene import SyntheticScene, GaussianModel

dev = torch.device("cuda:0")
scene = SyntheticScene("h36m", n_views=4, seed=0, device=dev)
gm = GaussianModel().create_from_points(scene.pose_3d_init, scene.spatial_lr_scale, scene.n_joints, scene_type="h36m", device=dev)
gm.training_setup()
hm = generate_heatmaps(gm._xyz.detach(), gm.get_scaling.detach(), gm._rotation.detach(), torch.tensor(scene.poses_2d, device=dev), scene.cameras)
render = render_functions["diff-gaussian-rasterization-h36m"]
pipe = types.SimpleNamespace(debug=False, antialiasing=False, compute_cov3D_python=False, convert_SHs_python=False)
bg = torch.zeros(3, device=dev)


def it(i):
    pkg = render(scene.cameras[i % 4], gm, pipe, bg)
    loss, _ = l2_loss_gaussian(pkg["render"], hm[i % 4])
    loss.backward()
    if (i + 1) % 4 == 0:
        gm.optimizer.step()
        gm.optimizer.zero_grad(set_to_none=True)


for i in range(16):
    it(i)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for i in range(200):
    it(i)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(45)
