"""One frame's 500 iterations as hipGraphs (MultiViewLoop.run) and one accumulation group, both datasets: used for
interleaved A/B of library builds (SKS_LIB_OVERRIDE)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from skelsplat_amd.heatmaps import generate_heatmaps
from skelsplat_amd.loop import MultiViewLoop
from skelsplat_amd.scene import SyntheticScene, GaussianModel

dev = torch.device("cuda:0")
for ds, V in (("h36m", 4), ("panoptic", 31)):
    sc = SyntheticScene(ds, n_views=V, seed=0, device=dev)
    gm = GaussianModel().create_from_points(sc.pose_3d_init, sc.spatial_lr_scale, sc.n_joints, scene_type=ds, device=dev)
    gm.training_setup()
    hm = generate_heatmaps(gm._xyz.detach(), gm.get_scaling.detach(), gm._rotation.detach(), torch.tensor(sc.poses_2d, device=dev), sc.cameras)
    loop = MultiViewLoop(gm, sc.cameras, hm, dataset=ds, accumulation_steps=V, use_graph=True)
    loop.run(500)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(5):
        loop.iteration = 0
        loop.run(500)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t) / 5 * 1e3
    print(f"{os.path.basename(os.environ.get('SKS_LIB_OVERRIDE', 'tree'))} {ds}: 500-iteration scene {ms:.3f} ms = {ms / (500 / V) * 1e3:.1f} us per {V}-view group", flush=True)
