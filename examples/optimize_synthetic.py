#!/usr/bin/env python3
"""End-to-end run of the hot path on one synthetic frame, shaped like the reference's train.py + eval.py:
noisy 2D detections -> DLT initial guess -> pseudo-GT heat-maps -> 500 iterations of the multi-view loop -> ply +
MPJPE.  python examples/optimize_synthetic.py [--dataset h36m|panoptic|occlusion-person] [--views V] [--iters 500]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from skelsplat_amd import io, triangulation
from skelsplat_amd.heatmaps import generate_heatmaps
from skelsplat_amd.loop import MultiViewLoop
from skelsplat_amd.scene import GaussianModel, SyntheticScene


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dataset", default="h36m")
    ap.add_argument("--views", type=int, default=4)
    ap.add_argument("--iters", type=int, default=500)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--init", default="noisy", choices=["noisy", "dlt"])
    ap.add_argument("--out", default="gpurun_out/example")
    ap.add_argument("--dense", action="store_true", help="dense render + fused loss instead of the sparse fused step")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    sc = SyntheticScene(args.dataset, n_views=args.views, seed=args.seed, device=dev)
    if args.init == "dlt":   # BASELINE config 1: linear triangulation of the noisy detections
        init = triangulation.triangulate_poses(triangulation.projection_matrices(sc.cameras), sc.poses_2d)[:, :3]
    else:
        init = sc.pose_3d_init
    gm = GaussianModel().create_from_points(init, sc.spatial_lr_scale, sc.n_joints, scene_type=args.dataset, device=dev)
    gm.training_setup()
    hm = generate_heatmaps(gm._xyz.detach(), gm.get_scaling.detach(), gm._rotation.detach(),
                           torch.tensor(sc.poses_2d, device=dev), sc.cameras)
    loop = MultiViewLoop(gm, sc.cameras, hm, dataset=args.dataset, accumulation_steps=args.views, sparse=not args.dense,
                         use_graph=True)
    e0 = io.mpjpe(init, sc.pose_3d_gt)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    loop.run(args.iters)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    # a second frame seen by the same cameras reuses the captured graphs (what a dataset-scale run sees per frame):
    # parameters, optimiser state, heat-maps and tile statistics are re-initialised in place
    result = [p.detach().clone() for p in (gm._xyz, gm._scaling, gm._rotation, gm._opacity)]
    gm2 = type("R", (), dict(_xyz=result[0], _scaling=result[1], _rotation=result[2], _opacity=result[3]))
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    loop.new_scene(init, poses_2d=sc.poses_2d)
    loop.run(args.iters)
    torch.cuda.synchronize()
    dt2 = time.perf_counter() - t1
    assert torch.allclose(gm._xyz, gm2._xyz, atol=1e-3), "replayed scene differs from the first run"
    pred = gm._xyz.detach().cpu().numpy()
    path = os.path.join(args.out, "point_cloud", f"iteration_{args.iters}", "synthetic_0.ply")
    io.save_ply(path, gm)
    assert np.allclose(io.read_ply_xyz(path), pred, atol=1e-4)
    S, N = loop.last_losses
    print(f"{args.dataset} V={args.views} {sc.W}x{sc.H}: {args.iters} iterations in {dt * 1e3:.1f} ms first scene, "
          f"{dt2 * 1e3:.1f} ms with graphs cached ({args.iters / dt2:.0f} it/s); MPJPE {e0:.2f} mm -> {io.mpjpe(pred, sc.pose_3d_gt):.2f} mm "
          f"(root-relative {io.mpjpe_root_relative(pred, sc.pose_3d_gt):.2f} mm); last losses {(S / N).tolist()}")


if __name__ == "__main__":
    main()
