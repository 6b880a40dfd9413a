"""Where does the forward kernel's fixed cost come from?  (a) normal scene (b) everything culled (pure fill)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from skelsplat_amd import _lib, rasterizer as R
from tools.tune_fwd import setup

def t_fwd(views, params, iters=30, tune=0):
    for _ in range(3):
        R.forward_views(views, *params, tune_flags=tune)
    torch.cuda.synchronize()
    _lib.prof_enable(True); _lib.prof_read(0)
    for _ in range(iters):
        R.forward_views(views, *params, tune_flags=tune)
    torch.cuda.synchronize()
    f, n = _lib.prof_read(0); _lib.prof_enable(False)
    return f / n * 1e3

for dataset, V in (("h36m", 4), ("panoptic", 8)):
    scene, views, params, dL = setup(dataset, V)
    print(dataset, V, "normal   ", round(t_fwd(views, params), 1), "us")
    far = list(params); far[0] = params[0] * 0 + torch.tensor([0., 0., 1e7], device=params[0].device)   # behind / outside every camera
    print(dataset, V, "culled   ", round(t_fwd(views, far), 1), "us")
    one = list(params); m = params[0].clone(); m[1:] = torch.tensor([0., 0., 1e7], device=m.device); one[0] = m
    print(dataset, V, "one joint", round(t_fwd(views, one), 1), "us")
    tiny = list(params); tiny[3] = params[3] * 0.05   # 1 mm splats: radius ~2 px
    print(dataset, V, "tiny     ", round(t_fwd(views, tiny), 1), "us")
