#!/usr/bin/env python3
"""A whole synthetic SEQUENCE through the hot path, the way the reference's train.py walks a dataset (one frame after the
other, train.py:74-99) -- but many frames per launch: noisy 2D detections of N frames -> DLT initial guesses -> 500
iterations of the multi-view loop for every frame (loop.FramePipeline) -> MPJPE per frame.
python examples/optimize_sequence.py [--frames 64] [--per-launch 16] [--streams 2] [--iters 500]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from skelsplat_amd import io, triangulation
from skelsplat_amd.loop import FramePipeline
from skelsplat_amd.scene import GaussianModel, SyntheticScene, project_points


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dataset", default="h36m")
    ap.add_argument("--views", type=int, default=4)
    ap.add_argument("--frames", type=int, default=64)
    ap.add_argument("--per-launch", type=int, default=16)
    ap.add_argument("--streams", type=int, default=2)
    ap.add_argument("--iters", type=int, default=500)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    sc = SyntheticScene(args.dataset, n_views=args.views, seed=0, device=dev)
    rng = np.random.default_rng(1)
    # a moving skeleton: the template drifts and wobbles from frame to frame; detections carry 3 px of noise
    gt = np.stack([sc.pose_3d_gt + np.array([8.0 * f, 3.0 * f, 0.0]) + rng.normal(0, 5.0, sc.pose_3d_gt.shape)
                   for f in range(args.frames)])
    p2d = np.stack([np.stack([project_points(c, gt[f]) + rng.normal(0, 3.0, (sc.n_points, 2)) for c in sc.cameras])
                    for f in range(args.frames)]).astype(np.float32)
    Pm = triangulation.projection_matrices(sc.cameras)
    init = np.stack([triangulation.triangulate_poses(Pm, p2d[f])[:, :3] for f in range(args.frames)]).astype(np.float32)
    gm = GaussianModel().create_from_points(init[0], sc.spatial_lr_scale, sc.n_joints, scene_type=args.dataset, device=dev)
    gm.training_setup()
    pipe = FramePipeline(gm, sc.cameras, frames=args.per_launch, streams=args.streams, dataset=args.dataset,
                         accumulation_steps=args.views)
    pipe.optimize_sequence(init, p2d, iterations=args.iters)          # captures the hipGraphs
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = pipe.optimize_sequence(init, p2d, iterations=args.iters)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    pred = out.cpu().numpy()
    e0 = np.mean([io.mpjpe(init[f], gt[f]) for f in range(args.frames)])
    e1 = np.mean([io.mpjpe(pred[f], gt[f]) for f in range(args.frames)])
    print(f"{args.dataset} V={args.views} {sc.W}x{sc.H}: {args.frames} frames x {args.iters} iterations in {dt * 1e3:.1f} ms "
          f"({args.frames / dt:.0f} frames/s, {args.per_launch} frames per launch on {args.streams} streams); "
          f"mean MPJPE {e0:.2f} mm (DLT) -> {e1:.2f} mm")


if __name__ == "__main__":
    main()
