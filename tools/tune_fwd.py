"""GPU tuning sweep for the fused forward / gather backward (prints kernel times from the hipEvent hooks)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from skelsplat_amd import _lib, rasterizer as R
from skelsplat_amd.scene import SyntheticScene, GaussianModel

def setup(dataset, V):
    dev = torch.device("cuda:0")
    scene = SyntheticScene(dataset, n_views=V, seed=0, device=dev)
    gm = GaussianModel().create_from_points(scene.pose_3d_init, scene.spatial_lr_scale, scene.n_joints, scene_type=dataset, device=dev)
    P, C = scene.n_points, scene.n_joints
    with torch.no_grad():
        params = (gm.get_xyz.detach().clone(), gm.get_features.reshape(P, C).contiguous(), gm.get_opacity.detach().clone(),
                  gm.get_scaling.detach().clone(), gm.get_rotation.detach().clone(), None)
    views = R.ViewBatch.from_cameras(scene.cameras)
    dL = torch.randn((V, C, scene.H, scene.W), device=dev)
    return scene, views, params, dL

def run(views, params, dL, tune, iters=30):
    for _ in range(3):
        c, i, r, st = R.forward_views(views, *params, tune_flags=tune)
        R.backward_views(st, *params, dL)
    torch.cuda.synchronize()
    _lib.prof_enable(True); _lib.prof_read(0); _lib.prof_read(1)
    t0 = time.perf_counter()
    for _ in range(iters):
        c, i, r, st = R.forward_views(views, *params, tune_flags=tune)
        R.backward_views(st, *params, dL)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    f, fn = _lib.prof_read(0); b, bn = _lib.prof_read(1)
    _lib.prof_enable(False)
    return f / fn * 1e3, b / bn * 1e3, dt * 1e6

if __name__ == "__main__":
  for dataset, V in (("h36m", 4), ("h36m", 16), ("panoptic", 8), ("panoptic", 31)):
      scene, views, params, dL = setup(dataset, V)
      nbytes = 4.0 * scene.H * scene.W * (scene.n_joints + 1) * V
      for pb in (1, 2, 3, 4, 6, 8, 16):
          f, b, tot = run(views, params, dL, pb << 8)
          print(f"{dataset} V={V} passes/block={pb}: fwd {f:7.1f} us ({nbytes/f/1e3:6.0f} GB/s)  bwd {b:6.1f} us  step wall {tot:7.1f} us", flush=True)

  # host-side enqueue cost (no sync inside): is the step CPU-bound?
  import time
  scene, views, params, dL = setup("h36m", 4)
  for _ in range(20):
      c, i, r, st = R.forward_views(views, *params); R.backward_views(st, *params, dL)
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for _ in range(200):
      c, i, r, st = R.forward_views(views, *params)
  t1 = time.perf_counter()
  torch.cuda.synchronize()
  t2 = time.perf_counter()
  for _ in range(200):
      R.backward_views(st, *params, dL)
  t3 = time.perf_counter()
  torch.cuda.synchronize()
  print(f"host enqueue: forward_views {1e6*(t1-t0)/200:.1f} us/call, backward_views {1e6*(t3-t2)/200:.1f} us/call")
