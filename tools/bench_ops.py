"""Timings of the smaller ops against their plain-PyTorch GPU counterparts (BASELINE.md §3 shapes)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from skelsplat_amd import ops
from tests.test_ops_gpu import ssim_torch

dev = torch.device("cuda:0")

def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3

for shape in ((5, 5, 1080, 1920), (1, 17, 1000, 1000), (5, 1, 1500, 1500)):
    a = torch.rand(shape, device=dev, requires_grad=True)
    b = torch.rand(shape, device=dev)
    n = a.numel() * 4
    def ours_train():
        a.grad = None
        ops.fused_ssim(a, b).backward()
    def torch_train():
        a.grad = None
        ssim_torch(a, b).mean().backward()
    with torch.no_grad():
        ti = timeit(lambda: ops.fused_ssim(a, b, train=False))
        tti = timeit(lambda: ssim_torch(a, b).mean())
    tt = timeit(ours_train)
    ttt = timeit(torch_train)
    print(f"fused_ssim {shape}: inference {ti:.3f} ms ({3*n/ti/1e6:.0f} GB/s alg: 2 reads + 1 write) vs torch conv2d {tti:.3f} ms ({tti/ti:.1f}x); "
          f"train fwd+bwd {tt:.3f} ms ({12*n/tt/1e6:.0f} GB/s alg) vs torch {ttt:.3f} ms ({ttt/tt:.1f}x)")

for P in (17, 4352, 100000, 1000000):
    pts = torch.randn(P, 3, device=dev) * 100
    t = timeit(lambda: ops.distCUDA2(pts))
    ta = timeit(lambda: ops.distCUDA2(pts, method="allpairs"), iters=3) if 2048 < P <= 200000 else (t if P <= 2048 else float("nan"))
    def brute():
        if P > 20000:
            return None
        d = torch.cdist(pts, pts) ** 2
        d.fill_diagonal_(float("inf"))
        return d.topk(3, largest=False).values.mean(1)
    tb = timeit(brute) if P <= 20000 else float("nan")
    print(f"distCUDA2 P={P}: {t:.3f} ms (all-pairs sweep {ta:.3f} ms) vs torch cdist+topk {tb:.3f} ms")

r = torch.rand((4, 17, 1000, 1000), device=dev) * (torch.rand((4, 17, 1000, 1000), device=dev) > 0.9)
g = torch.rand((4, 17, 1000, 1000), device=dev) * (torch.rand((4, 17, 1000, 1000), device=dev) > 0.9)
n = r.numel() * 4
t = timeit(lambda: ops.masked_l2(r, g))
def torch_l2():
    rr = r.clone().requires_grad_(True)
    mask = (g > 0) | (rr > 0)
    loss = sum(((rr[v] - g[v]) ** 2)[mask[v]].mean() for v in range(4))
    loss.backward()
tt = timeit(torch_l2, iters=5)
print(f"masked_l2 (4,17,1000,1000): {t:.3f} ms ({3*n/t/1e6:.0f} GB/s alg: 2 reads + 1 write) vs torch ops + autograd {tt:.3f} ms ({tt/t:.1f}x)")

# per-scene preparation of the sparse fused loop: closed-form heat-maps + their per-tile statistics
from skelsplat_amd import rasterizer as R
from skelsplat_amd.scene import SyntheticScene, GaussianModel
from skelsplat_amd.heatmaps import generate_heatmaps
for ds, V in (("h36m", 4), ("panoptic", 31)):
    sc = SyntheticScene(ds, n_views=V, seed=0, device=dev)
    gm = GaussianModel().create_from_points(sc.pose_3d_init, sc.spatial_lr_scale, sc.n_joints, scene_type=ds, device=dev)
    p2d = torch.tensor(sc.poses_2d, device=dev)
    mk = lambda: generate_heatmaps(gm._xyz.detach(), gm.get_scaling.detach(), gm._rotation.detach(), p2d, sc.cameras)
    hm = mk()
    hm = hm if torch.is_tensor(hm) else torch.stack(list(hm))
    th = timeit(mk, iters=5)
    ts = timeit(lambda: R.gt_tile_stats(hm), iters=10)
    nb = hm.numel() * 4
    print(f"{ds} V={V}: generate_heatmaps {th:.3f} ms ({nb/th/1e6:.0f} GB/s written), gt_tile_stats {ts:.3f} ms ({nb/ts/1e6:.0f} GB/s read)")
    del hm
