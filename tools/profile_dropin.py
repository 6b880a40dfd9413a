"""Where the literal drop-in iteration (render() -> criterion -> backward) spends its time: host wall clock of each
part with and without a device sync, plus the top ops of a torch profiler trace."""
import os
import sys
import time
import types

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gaussian_renderer import render_functions
from skelsplat_amd.heatmaps import generate_heatmaps
if "--tensor" in sys.argv:     # the reference's own tensor-op criterion (loss_utils.py:86-100 as restated in loop.py) instead of the fused one
    from skelsplat_amd.loop import l2_loss_gaussian
else:
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


def it(i, sync=False):
    t = [time.perf_counter()]
    pkg = render(scene.cameras[i % 4], gm, pipe, bg)
    if sync: torch.cuda.synchronize()
    t.append(time.perf_counter())
    loss, _ = l2_loss_gaussian(pkg["render"], hm[i % 4])
    if sync: torch.cuda.synchronize()
    t.append(time.perf_counter())
    loss.backward()
    if sync: torch.cuda.synchronize()
    t.append(time.perf_counter())
    if (i + 1) % 4 == 0:
        gm.optimizer.step()
        gm.optimizer.zero_grad(set_to_none=True)
    if sync: torch.cuda.synchronize()
    t.append(time.perf_counter())
    return [1e6 * (b - a) for a, b in zip(t, t[1:])]


for sync in (False, True):
    for i in range(8):
        it(i, sync)
    torch.cuda.synchronize()
    acc = [0.0] * 4
    n = 40
    t0 = time.perf_counter()
    for i in range(n):
        acc = [a + b for a, b in zip(acc, it(i, sync))]
    torch.cuda.synchronize()
    tot = 1e6 * (time.perf_counter() - t0) / n
    print(f"sync={sync}: render {acc[0]/n:.0f} us, criterion {acc[1]/n:.0f} us, backward {acc[2]/n:.0f} us, optimizer {acc[3]/n:.0f} us; iteration {tot:.0f} us")

if "--no-torch-profiler" in sys.argv:
    sys.exit(0)
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for i in range(8):
        it(i)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=22, max_name_column_width=48))
print(prof.key_averages().table(sort_by="self_cuda_time_total", row_limit=22, max_name_column_width=60))
