#!/usr/bin/env python3
"""Timing of the small-path kernels across P: one skeleton (wave-resident backward, P <= 64), 4 and 12 skeletons
(P = 68, 204: LDS-list gather backward), 4 views at 1000x1000.  Informational."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import util
from skelsplat_amd import rasterizer as R

dev = torch.device("cuda", 0)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev) if a is not None else None
V, C, H, W = 4, 17, 1000, 1000
for nsk in (1, 4, 12):
    case = util.make_case(seed=7, W=W, H=H, n_views=V, scale_log=3.0, n_skeletons=nsk, pitch=900.0, ring=5000.0 + 600.0 * nsk, onehot=True, opac=1.0)
    views = R.ViewBatch.from_cameras([cam.to(dev) for cam in case.cams])
    args = (t(case.means), t(case.feat), t(case.opac), t(case.scales), t(case.quats), None)
    dL = torch.randn((V, C, H, W), device=dev)
    color, inv, radii, st = R.forward_views(views, *args)
    for name, fn in (("forward", lambda: R.forward_views(views, *args)), ("backward", lambda: R.backward_views(st, *args, dL))):
        for _ in range(5): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(50): fn()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 50
        print(f"P={case.P:4d} visible/view {int((radii > 0).sum()) / V:.0f}  {name}: {dt * 1e6:.1f} us per {V}-view call", flush=True)
