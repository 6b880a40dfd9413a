import os, sys
sys.path.insert(0, os.getcwd())
import torch, bench
from skelsplat_amd import rasterizer as R, _lib
dev = torch.device("cuda", 0)
wl = bench.WORKLOADS["h36m"]
scene, gm, params = bench.make_scene(torch, wl, dev)
views = R.ViewBatch.from_cameras(scene.cameras)
dL = torch.randn((4, 17, scene.H, scene.W), device=dev)
step = bench.ApiStep(views, params, dL)
for _ in range(30): step()
torch.cuda.synchronize()
for rec in (False, True, False, True):
    _lib.prof_enable(True, every=1, kinds=(0,), recorded=rec)
    _lib.prof_read(0)
    for _ in range(64): step()
    torch.cuda.synchronize()
    ms, n, q = _lib.prof_read_quantiles(0)
    print("recorded" if rec else "ext", f"avg {1e3*ms/n:.2f} us p10/p50/p90 {[round(1e3*x,2) for x in q]}")
_lib.prof_enable(False)
