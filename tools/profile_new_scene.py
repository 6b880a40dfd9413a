"""Where a streamed frame's preparation time goes (MultiViewLoop.new_scene), wall clock with syncs around each part."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from skelsplat_amd import rasterizer as R
from skelsplat_amd.heatmaps import generate_heatmaps, heatmap_factors
from skelsplat_amd.loop import MultiViewLoop
from skelsplat_amd.scene import SyntheticScene, GaussianModel

ds = sys.argv[1] if len(sys.argv) > 1 else "h36m"
V = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dev = torch.device("cuda:0")
scene = SyntheticScene(ds, n_views=V, seed=0, device=dev)
gm = GaussianModel().create_from_points(scene.pose_3d_init, scene.spatial_lr_scale, scene.n_joints, scene_type=ds, device=dev)
gm.training_setup()
p2d = torch.tensor(scene.poses_2d, device=dev)
pts = torch.tensor(scene.pose_3d_init, device=dev, dtype=torch.float32)
hm = generate_heatmaps(gm._xyz.detach(), gm.get_scaling.detach(), gm._rotation.detach(), p2d, scene.cameras)
loop = MultiViewLoop(gm, scene.cameras, hm, dataset=ds, accumulation_steps=V, use_graph=True)
loop.run(500)


def t(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


args = (gm._xyz.detach(), gm.get_scaling.detach(), gm._rotation.detach(), p2d, scene.cameras)
print(f"{ds} V={V}")
print(f"  reset_from_points        {t(lambda: gm.reset_from_points(pts)):.3f} ms")
print(f"  heatmap_factors          {t(lambda: heatmap_factors(*args)):.3f} ms")
print(f"  generate_heatmaps(out=)  {t(lambda: generate_heatmaps(*args, out=hm)):.3f} ms")
print(f"  gt_tile_stats            {t(lambda: R.gt_tile_stats(hm)):.3f} ms")
print(f"  new_scene                {t(lambda: loop.new_scene(pts, poses_2d=p2d)):.3f} ms")
print(f"  run(500)                 {t(lambda: (setattr(loop, 'iteration', 0), loop.run(500)), n=5):.3f} ms")
print(f"  new_scene + run(500)     {t(lambda: (loop.new_scene(pts, poses_2d=p2d), loop.run(500)), n=5):.3f} ms")
