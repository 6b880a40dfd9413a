#!/usr/bin/env python3
"""Frames per second of the H36M training loop (4 views @ 1000x1000, 500 iterations per frame, heat-map generation
included) when F independent frames share the launches (loop.FrameBatchLoop), against one frame at a time
(loop.MultiViewLoop.new_scene + run, hipGraphs in both).  Usage: bench_frames.py [F ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from skelsplat_amd.loop import MultiViewLoop, FrameBatchLoop
from skelsplat_amd.scene import SyntheticScene, GaussianModel

dev = torch.device("cuda", 0)
Fs = [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8, 16]
ITERS = int(os.environ.get("ITERS", "500"))
FACTORED = os.environ.get("FACTORED", "1") == "1"   # heat-maps as separable factors (no planes) or as planes
sc = SyntheticScene("h36m", n_views=4, seed=0, device=dev)
rng = np.random.default_rng(1)
base3, base2 = np.asarray(sc.pose_3d_init, np.float32), np.asarray(sc.poses_2d, np.float32)


def model():
    gm = GaussianModel().create_from_points(sc.pose_3d_init, sc.spatial_lr_scale, sc.n_joints, device=dev)
    gm.training_setup()
    return gm


def frames(F):
    return (np.stack([base3 + rng.normal(0, 10.0, base3.shape) for _ in range(F)]).astype(np.float32),
            np.stack([base2 + rng.normal(0, 2.0, base2.shape) for _ in range(F)]).astype(np.float32))


# one frame at a time
base = None
if os.environ.get("ONLY_BATCH") != "1":
  gm = model()
  hm = torch.zeros((4, sc.n_joints, sc.H, sc.W), device=dev)
  one = MultiViewLoop(gm, sc.cameras, hm, dataset="h36m", sparse=True, use_graph=True)
  pts, p2d = frames(8)
  for rep in range(2):
      torch.cuda.synchronize(); t0 = time.perf_counter()
      for f in range(8):
          one.new_scene(pts[f], poses_2d=p2d[f])
          one.run(ITERS)
      torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 8
  print(f"one frame at a time: {dt*1e3:.3f} ms per frame, {1/dt:.0f} frames/s")
  base = dt
for F in Fs:
    fb = FrameBatchLoop(model(), sc.cameras, F, dataset="h36m", use_graph=True, factored=FACTORED)
    pts, p2d = frames(F)
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        fb.new_scenes(pts, poses_2d=p2d)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        fb.run(ITERS)
        torch.cuda.synchronize(); t2 = time.perf_counter()
    dt = (t2 - t0) / F
    print(f"F = {F:2d}: {(t2-t0)*1e3:.3f} ms per batch (heat-maps {(t1-t0)*1e3:.3f} ms, loop {(t2-t1)*1e3:.3f} ms = "
          f"{(t2-t1)*1e6/(ITERS/4):.1f} us per group), {1/dt:.0f} frames/s, x{(base or dt)/dt:.2f}")
    del fb
    torch.cuda.empty_cache()

# several batches on as many streams: one's single-workgroup tails run under the others' backward kernels
for spec in os.environ.get("STREAMS", "2x8,2x16,4x8,4x16").split(","):
    if not spec:
        continue
    ns, h = (int(x) for x in spec.split("x"))
    F = ns * h
    loops = [FrameBatchLoop(model(), sc.cameras, h, dataset="h36m", use_graph=True, factored=FACTORED) for _ in range(ns)]
    streams = [torch.cuda.Stream() for _ in range(ns)]
    pts, p2d = frames(F)
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i, (fb, st) in enumerate(zip(loops, streams)):
            with torch.cuda.stream(st):
                fb.new_scenes(pts[i * h:(i + 1) * h], poses_2d=p2d[i * h:(i + 1) * h])
        for k in range(0, ITERS, 100):      # interleave the graph launches of the streams
            for fb, st in zip(loops, streams):
                with torch.cuda.stream(st):
                    fb.run(min(ITERS, k + 100))
        torch.cuda.synchronize(); t2 = time.perf_counter()
    dt = (t2 - t0) / F
    print(f"{ns} streams x {h:2d} frames: {(t2-t0)*1e3:.3f} ms per {F} frames, {1/dt:.0f} frames/s")
    del loops
    torch.cuda.empty_cache()
