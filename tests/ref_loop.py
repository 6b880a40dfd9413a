"""Literal per-iteration restatement of the reference loop (train.py:130-222) on top of the PyTorch oracle
(oracle/torch_ref.py + autograd).  TEST INFRASTRUCTURE: the thing the HIP loop is compared against."""
import math

import torch

from oracle import torch_ref
from skelsplat_amd.loop import limb_3d_consistency_loss, l2_loss_gaussian


def render_ref(cam, gm, W, H):
    """gaussian_renderer.render_* (reference :28-138) with the oracle rasterizer; returns the clamped image."""
    P = gm._xyz.shape[0]
    color, radii, inv = torch_ref.rasterize(
        gm.get_xyz, None, gm.get_features.reshape(P, -1), gm.get_opacity, gm.get_scaling, gm.get_rotation, None,
        cam.world_view_transform, cam.full_proj_transform, W, H, math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5))
    return color.clamp(0, 1)


def view_grads_ref(gm, cam, gt, W, H, dataset, lambda_consistency):
    """One iteration's loss and autograd.grad wrt (xyz, _scaling, _rotation, _opacity) (train.py:140-161)."""
    image = render_ref(cam, gm, W, H)
    l2, _ = l2_loss_gaussian(image, gt)
    loss = l2 + limb_3d_consistency_loss(gm.get_xyz, dataset) * lambda_consistency
    params = [gm._xyz, gm._scaling, gm._rotation, gm._opacity]
    grads = torch.autograd.grad(loss, params, allow_unused=True)
    grads = [g if g is not None else torch.zeros_like(p) for g, p in zip(grads, params)]
    return loss.detach(), grads


def run_reference_loop(gm, cameras, heatmaps, W, H, dataset, iterations, accumulation_steps=4, lambda_consistency=1e-5,
                       on_step=None, early_stopping=None):
    """`on_step(gm)`: called after every optimiser step.  `early_stopping`: a callable loss -> bool fed `loss.item()` of every
    iteration like train.py:155; when it fires the optimiser steps at once and the loop ends (train.py:182, 227-233) -- the
    stopping iteration is then left in `run_reference_loop.stopped_at` (None otherwise).  (This restatement is itself held to
    the reference's own train.training(), run in the build container: tests/golden/reference_loop.npz, tests/test_loop_golden.py.)"""
    V = len(cameras)
    accumulated = torch.zeros((V,) + tuple(gm._xyz.shape))
    cam_idx_counter = 0
    run_reference_loop.stopped_at = None
    for iteration in range(1, iterations + 1):
        gm.update_learning_rate(iteration)
        idx = cam_idx_counter % V
        cam_idx_counter += 1
        loss, (gx, gs, gr, go) = view_grads_ref(gm, cameras[idx], heatmaps[idx], W, H, dataset, lambda_consistency)
        stop = early_stopping is not None and bool(early_stopping(loss.item()))
        accumulated[idx] = gx
        gm._scaling.grad, gm._rotation.grad, gm._opacity.grad = gs, gr, go
        if iteration % accumulation_steps == 0 or stop:
            gm._xyz.grad = accumulated.mean(dim=0)
            with torch.no_grad():
                gm.optimizer.step()
                gm.optimizer.zero_grad(set_to_none=True)
            if on_step is not None:
                on_step(gm)
        if stop:
            run_reference_loop.stopped_at = iteration
            break
    return gm._xyz.detach().clone()
