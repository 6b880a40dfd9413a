import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from skelsplat_amd import rasterizer as R
from skelsplat_amd.scene import stress_scene
dev = torch.device("cuda", 0)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev) if a is not None else None
sc, g = stress_scene(8)
views = R.ViewBatch.from_cameras([cam.to(dev) for cam in sc.cameras])
args = (t(g["means"]), t(g["feat"]), t(g["opac"]), t(g["scales"]), t(g["quats"]), None)
color, inv, radii, st = R.forward_views(views, *args, bin_capacity=400000)
pl, rg, nr = R.export_lists(st)
n = (rg[..., 1] - rg[..., 0]).cpu().numpy().reshape(-1)
n = n[n > 0]
print("non-empty", len(n), "mean", n.mean(), "max", n.max(), "hist", np.bincount(n)[:40].tolist())
for G in (2048, 4096):
    for run in (1, 4):
        idx = np.arange(len(n))
        blk = (idx // run) % G
        s = np.bincount(blk, weights=n, minlength=G)
        print("blocks", G, "run", run, "per-block entries: mean %.1f max %.0f min %.0f" % (s.mean(), s.max(), s.min()))
