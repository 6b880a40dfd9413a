#!/usr/bin/env python3
"""Timing of BASELINE config 5 (stress: 256 skeletons, P = 4352, C = 17, V = 8 views at 2048x2048) on the binned path.
Informational -- bench.py's headline stays the H36M configuration."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from skelsplat_amd import rasterizer as R, _lib
from skelsplat_amd.scene import stress_scene

dev = torch.device("cuda", 0)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev) if a is not None else None
V = int(os.environ.get("V", "8"))
sc, g = stress_scene(V)
P = sc.n_points
views = R.ViewBatch.from_cameras([cam.to(dev) for cam in sc.cameras])
args = (t(g["means"]), t(g["feat"]), t(g["opac"]), t(g["scales"]), t(g["quats"]), None)
C, H, W = 17, 2048, 2048
dL = torch.randn((V, C, H, W), device=dev)
color, inv, radii, st = R.forward_views(views, *args, bin_capacity=400000, check_capacity=True)
print("P", P, "num_rendered per view", st.num_rendered_dev[:V].cpu().tolist(), "visible", int((radii > 0).sum()) / V)
ws = R.Workspace()     # what a loop does: the same buffers every step (no allocation, no clearing launch: SKS_BIN_CLEAN)
fwd = lambda: R.forward_views(views, *args, bin_capacity=400000, workspace=ws, check_capacity="auto")
st = fwd()[3]
for name, fn in (("forward", fwd),
                 ("backward", lambda: R.backward_views(st, *args, dL, workspace=ws)),
                 ("fwd+bwd", lambda: R.backward_views(fwd()[3], *args, dL, workspace=ws))):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 20
    for _ in range(n): fn()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    alg = 4.0 * H * W * (C + 1) * V * (2 if name == "fwd+bwd" else 1)
    print(f"{name}: {dt*1e3:.3f} ms per {V}-view call, {V/dt:.0f} views/s, algorithmic {alg/dt/1e9:.0f} GB/s")

# dispatch floor of the tile-per-block kernels: every Gaussian culled -> every tile empty.  Only on request: under
# rocprofv3 these launches would be averaged into the same kernel names as the real ones (round 1's kernel-stats file
# showed k_render_bwd_binned "43 .. 337 us" for exactly that reason: 43 = all tiles empty, 337 = the stress scene)
if "--floor" not in sys.argv:
    sys.exit(0)
far = (torch.tensor([[0.0, 0.0, 1e9]], device=dev).repeat(P, 1),) + args[1:]
c2, i2, r2, st2 = R.forward_views(views, *far, force_binned=True, bin_capacity=400000)
for name, fn in (("forward, all tiles empty", lambda: R.forward_views(views, *far, force_binned=True, bin_capacity=400000, check_capacity="auto")),
                 ("backward, all tiles empty", lambda: R.backward_views(st2, *far, dL))):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): fn()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    print(f"{name}: {dt*1e3:.3f} ms")
nonempty = (st.binning is not None)
pl, rg, nr = R.export_lists(st)
ne = (rg[..., 1] > rg[..., 0]).sum(dim=1)
print("non-empty tiles per view", ne.cpu().tolist(), "of", rg.shape[1])
