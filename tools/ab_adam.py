#!/usr/bin/env python3
"""Drop-in iteration (bench.py dropin_iteration, fused criterion) with torch.optim.Adam as the reference builds it (foreach) against
fused=True, the single-tensor path and skelsplat_amd.optim.Adam (one launch): ms per view, host time of optimizer.step(), and how far the parameters drift apart over 64 steps."""
import os, sys, time, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from gaussian_renderer import render_functions
from skelsplat_amd.ops import l2_loss_gaussian as crit
from skelsplat_amd.heatmaps import generate_heatmaps

dev = torch.device("cuda", 0)
wl = bench.WORKLOADS["h36m"]
scene, gm, params = bench.make_scene(torch, wl, dev)
V = wl["V"]
render = render_functions["diff-gaussian-rasterization-h36m"]
pipe = types.SimpleNamespace(debug=False, antialiasing=False, compute_cov3D_python=False, convert_SHs_python=False)
bgc = torch.zeros(3, device=dev)


def run(fused, n=64 * V):
    gm2 = bench.fresh_model(scene, wl["dataset"], dev)
    if fused is not None:
        groups = [{k: v for k, v in g.items() if k in ("params", "lr", "name")} for g in gm2.optimizer.param_groups]
        if fused.get("ours"):
            from skelsplat_amd.optim import Adam
            gm2.optimizer = Adam(groups, lr=0.0, eps=1e-15)
        else:
            gm2.optimizer = torch.optim.Adam(groups, lr=0.0, eps=1e-15, **fused)
    hm = generate_heatmaps(gm2._xyz.detach(), gm2.get_scaling.detach(), gm2._rotation.detach(),
                           torch.tensor(scene.poses_2d, device=dev), scene.cameras)
    t_opt = [0.0]

    def it(i):
        pkg = render(scene.cameras[i % V], gm2, pipe, bgc)
        loss, _ = crit(pkg["render"], hm[i % V])
        loss.backward()
        if (i + 1) % V == 0:
            t0 = time.perf_counter()
            gm2.optimizer.step()
            gm2.optimizer.zero_grad(set_to_none=True)
            t_opt[0] += time.perf_counter() - t0
    for i in range(2 * V):
        it(i)
    reps = []
    for _ in range(3):
        t_opt[0] = 0.0
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(n):
            it(i)
        torch.cuda.synchronize()
        reps.append((1e3 * (time.perf_counter() - t0) / n, 1e6 * t_opt[0] / (n // V)))
    return sorted(reps)[1], gm2


for _ in range(2):
    for tag, kw in (("foreach (default)", None), ("fused=True", dict(fused=True)), ("foreach=False", dict(foreach=False)),
                    ("skelsplat_amd.optim.Adam", dict(ours=True))):
        (ms, us), g = run(kw)
        print(f"{tag:20s} {ms:.4f} ms per view, optimizer.step + zero_grad {us:.0f} us per step", flush=True)
        if kw is None:
            base = g
        else:
            d = max(float((a - b).abs().max()) for a, b in ((g._xyz, base._xyz), (g._scaling, base._scaling), (g._rotation, base._rotation), (g._opacity, base._opacity)))
            print(f"    max |parameter - default's| after {64 * 3 + 2} steps: {d:.3e} (xyz scale {float(base._xyz.abs().max()):.1f})")
