#!/usr/bin/env python3
"""Robustness soak: N frames streamed through ONE MultiViewLoop (new_scene + 500 iterations, hipGraphs) and through a FrameBatchLoop
(16 frames per launch); device memory must not grow, every result must be finite.   python tools/soak_scenes.py [frames]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from skelsplat_amd.loop import MultiViewLoop, FrameBatchLoop
from skelsplat_amd.heatmaps import generate_heatmaps

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
dev = torch.device("cuda", 0)
wl = bench.WORKLOADS["h36m"]
scene, gm, params = bench.make_scene(torch, wl, dev)
gm.training_setup()
p2d = torch.tensor(scene.poses_2d, device=dev)
hm = generate_heatmaps(gm._xyz.detach(), gm.get_scaling.detach(), gm._rotation.detach(), p2d, scene.cameras)
loop = MultiViewLoop(bench.fresh_model(scene, "h36m", dev), scene.cameras, hm, dataset="h36m", accumulation_steps=4, use_graph=True)
pts = torch.tensor(scene.pose_3d_init, device=dev, dtype=torch.float32)
g = torch.Generator(device=dev).manual_seed(0)
mem0 = None
t0 = time.time()
for i in range(N):
    loop.new_scene(pts + torch.randn(pts.shape, device=dev, generator=g) * 20.0, poses_2d=p2d + torch.randn(p2d.shape, device=dev, generator=g))
    out = loop.run(500)
    if i == 20:
        torch.cuda.synchronize(); mem0 = torch.cuda.memory_allocated()
    if i % 200 == 199:
        torch.cuda.synchronize()
        assert torch.isfinite(out).all()
        print(f"frame {i + 1}: {1e3 * (time.time() - t0) / (i + 1):.2f} ms per frame, allocated {torch.cuda.memory_allocated() / 2**20:.1f} MiB", flush=True)
torch.cuda.synchronize()
assert torch.cuda.memory_allocated() <= mem0 + (1 << 20), (torch.cuda.memory_allocated(), mem0)
fb = FrameBatchLoop(bench.fresh_model(scene, "h36m", dev), scene.cameras, frames=16, dataset="h36m", accumulation_steps=4, use_graph=True)
ptsN, p2dN = pts[None].repeat(16, 1, 1), p2d[None].repeat(16, 1, 1, 1)
memb = None
for i in range(max(2, N // 16)):
    fb.new_scenes(ptsN + torch.randn(ptsN.shape, device=dev, generator=g) * 20.0, poses_2d=p2dN)
    out = fb.run(500)
    if i == 3:
        torch.cuda.synchronize(); memb = torch.cuda.memory_allocated()
torch.cuda.synchronize()
assert torch.isfinite(out).all() and (memb is None or torch.cuda.memory_allocated() <= memb + (1 << 20))
print(f"ok: {N} frames one at a time, {max(2, N // 16) * 16} in batches of 16; {time.time() - t0:.0f} s")
