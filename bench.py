#!/usr/bin/env python3
"""bench.py -- rendered+backpropagated views/s of the skeletal-Gaussian rasterizer hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run, one rank per GPU)

A "step" is one accumulation group of the reference loop (train.py:130-222, accumulation_steps = V = 4): the V views
that share the Gaussian parameters are rendered (forward) and back-propagated (backward) with the upstream
gradient dL/d(render) already resident in HBM, followed for N > 1 by the exchange of per-view joint gradients
(all_gather over RCCL) that the view-sharded loop needs.  Work per GPU is fixed (V views) -> weak scaling.
Workload at N = 1: BASELINE.json configs[1], H36M 17 joints, 4 views @ 1000x1000 (synthetic skeleton + cameras).

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel (forward compositor): algorithmic bytes per
launch = 4*H*W*(C+1)*V (dense colour + inverse-depth planes it must write) / its average launch
duration measured with hipEvents on the launch stream.  `cpu_baseline` times the pure-PyTorch restatement
(oracle/torch_ref.py) on the host cores for a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6.3 TB/s is what a copy kernel achieves

WORKLOADS = {
    "h36m": dict(dataset="h36m", V=4, name="h36m_4view_1000x1000_P17_C17"),
    "panoptic": dict(dataset="panoptic", V=31, name="panoptic_31view_1920x1080_P19_C19"),
}


def cpu_baseline(scene, params, n_views):
    """Pure-PyTorch CPU rasterization (fwd + autograd bwd) of the first `n_views` views of the same scene."""
    import math
    import torch
    from oracle import torch_ref
    torch.set_num_threads(min(32, os.cpu_count() or 1))  # small ops: more threads only add sync overhead
    means, feat, opac, scales, quats = [p.detach().cpu() for p in params]
    W, H = scene.W, scene.H
    g = torch.Generator().manual_seed(0)

    def one(cam):
        m = means.clone().requires_grad_(True)
        s = scales.clone().requires_grad_(True)
        q = quats.clone().requires_grad_(True)
        o = opac.clone().requires_grad_(True)
        col, _, inv = torch_ref.rasterize(m, None, feat, o, s, q, None, cam.world_view_transform.cpu(),
                                          cam.full_proj_transform.cpu(), W, H, math.tan(cam.FoVx * 0.5),
                                          math.tan(cam.FoVy * 0.5))
        dL = torch.randn(col.shape, generator=g)
        (col * dL).sum().backward()
        return m.grad

    one(scene.cameras[0])  # warm-up
    t0 = time.perf_counter()
    done = 0
    while done < n_views or time.perf_counter() - t0 < 12.0:   # a bounded sample: ~12 s of CPU work (cap 25 s)
        one(scene.cameras[done % len(scene.cameras)])
        done += 1
        if time.perf_counter() - t0 > 25.0:
            break
    dt = time.perf_counter() - t0
    return dict(value=done / dt, unit="views/s", cores=torch.get_num_threads(), kind="port",
                sample=f"{done} views fwd+bwd of the same scene, oracle/torch_ref.py (pure PyTorch, fp32), {dt:.1f}s")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="h36m", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-prof", action="store_true", help="do not bracket kernels with hipEvents")
    ap.add_argument("--no-dropin", action="store_true", help="skip the literal drop-in iteration extra (its single-view "
                    "launches would mix into per-kernel averages of a rocprofv3 run)")
    args = ap.parse_args()

    # Native libraries write to the process's stdout too (RCCL prints a version banner from C stdio, flushed at exit, i.e.
    # AFTER the result line): fd 1 is pointed at stderr for the whole run and the one JSON line goes to the real stdout.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist
    from skelsplat_amd import _lib
    from skelsplat_amd import rasterizer as R
    from skelsplat_amd.scene import SyntheticScene, GaussianModel

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm device (no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or os.environ.get("SKS_BENCH_FORCE_DIST") == "1"   # the env switch exercises RCCL at world 1
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=dev)
    assert world == args.gpus or world == 1, (world, args.gpus)

    wl = WORKLOADS[args.workload]
    V = wl["V"]
    # every rank renders V views of the same skeleton from its own cameras (V * world views of one frame)
    scene = SyntheticScene(wl["dataset"], n_views=V, seed=rank, device=dev)
    ref_scene = SyntheticScene(wl["dataset"], n_views=V, seed=0)
    gm = GaussianModel().create_from_points(ref_scene.pose_3d_init, scene.spatial_lr_scale, scene.n_joints,
                                            scene_type=wl["dataset"], device=dev)
    W, H, P, C = scene.W, scene.H, scene.n_points, scene.n_joints
    views = R.ViewBatch.from_cameras(scene.cameras)
    with torch.no_grad():
        params = (gm.get_xyz.detach().clone(), gm.get_features.reshape(P, C).contiguous(), gm.get_opacity.detach().clone(),
                  gm.get_scaling.detach().clone(), gm.get_rotation.detach().clone())
    means, feat, opac, scales, quats = params
    dL = torch.randn((V, C, H, W), device=dev, generator=torch.Generator(device=dev).manual_seed(rank))
    gathered = torch.empty((world * V, P, 3), device=dev) if use_dist else None

    def step():
        color, inv, radii, st = R.forward_views(views, means, feat, opac, scales, quats, None)
        g = R.backward_views(st, means, feat, opac, scales, quats, None, dL)
        gx = g["means3D"]
        if use_dist:  # view-sharded loop: every rank needs all per-view joint gradients (train.py:175,215-217)
            dist.all_gather_into_tensor(gathered, gx)
            gx = gathered
        return gx.mean(dim=0)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    prof = not args.no_prof
    if prof:
        # the kernels of every 8th step are bracketed with hipEvents on the launch stream: >= 25 samples of the timed
        # region at the default 200 steps, without the ~12 us per step that four event records per step would add
        _lib.prof_enable(True, every=max(1, min(8, args.steps // 8)))
        _lib.prof_read(0), _lib.prof_read(1)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    fwd_ms = fwd_n = bwd_ms = bwd_n = 0
    fwd_q = bwd_q = (0.0, 0.0, 0.0)
    if prof:
        fwd_ms, fwd_n, fwd_q = _lib.prof_read_quantiles(0)
        bwd_ms, bwd_n, bwd_q = _lib.prof_read_quantiles(1)
        _lib.prof_enable(False)
    if use_dist:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    assert torch.isfinite(out).all()

    # ---- extras (not part of `value`): hipGraph replay of the same step, and the full loop step ----------------
    extras = {}
    if world == 1 and not use_dist:
        try:
            graph = torch.cuda.CUDAGraph()
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(3):
                    step()
            torch.cuda.current_stream().wait_stream(side)
            with torch.cuda.graph(graph):
                gout = step()
            for _ in range(5):
                graph.replay()
            torch.cuda.synchronize()
            tg = time.perf_counter()
            for _ in range(args.steps):
                graph.replay()
            torch.cuda.synchronize()
            tg = time.perf_counter() - tg
            extras["graph_replay_views_per_s"] = V * args.steps / tg
            extras["graph_replay_ms_per_step"] = 1e3 * tg / args.steps
        except Exception as e:  # capture is an optimisation, never a requirement
            extras["graph_replay_error"] = repr(e)[:200]
        try:
            from skelsplat_amd.loop import MultiViewLoop
            from skelsplat_amd.heatmaps import generate_heatmaps
            gm.training_setup()
            hm = generate_heatmaps(gm._xyz.detach(), gm.get_scaling.detach(), gm._rotation.detach(),
                                   torch.tensor(ref_scene.poses_2d, device=dev), scene.cameras)
            for tag, ug in (("", False), ("_hipgraph", True)):
                loop = MultiViewLoop(gm, scene.cameras, hm, dataset=wl["dataset"], accumulation_steps=V, use_graph=ug)
                for _ in range(5):
                    loop.step_group()
                torch.cuda.synchronize()
                tl = time.perf_counter()
                nl = max(10, args.steps // 2)
                for _ in range(nl):
                    loop.step_group()
                torch.cuda.synchronize()
                tl = time.perf_counter() - tl
                # V views rendered + fused masked-L2 + backward + device-side Adam step (train.py:130-222)
                extras["grad_step_ms" + tag] = 1e3 * tl / nl
                extras["loop_views_per_s" + tag] = V * nl / tl
            # one whole scene like configs/h36m.yaml (500 iterations), accumulation groups captured 25 per hipGraph
            loop = MultiViewLoop(gm, scene.cameras, hm, dataset=wl["dataset"], accumulation_steps=V, use_graph=True)
            loop.run(500)                      # captures
            torch.cuda.synchronize()
            ts = time.perf_counter()
            for _ in range(3):
                loop.iteration = 0             # parameters keep evolving; the work per iteration does not change
                loop.run(500)
            torch.cuda.synchronize()
            ts = (time.perf_counter() - ts) / 3
            extras["scene_500it_ms_hipgraph"] = 1e3 * ts
            extras["loop_views_per_s_scene_hipgraph"] = 500 / ts
            extras["grad_step_ms_scene_hipgraph"] = 1e3 * ts / (500 / V)
            # frames streamed through one loop object (train.py:74-99 per frame: re-initialise the Gaussians, generate the
            # heat-maps from the 2D detections, 500 iterations), everything in place so the hipGraphs are reused
            p2d = torch.tensor(ref_scene.poses_2d, device=dev)
            pts = torch.tensor(ref_scene.pose_3d_init, device=dev, dtype=torch.float32)
            loop.new_scene(pts, poses_2d=p2d)
            loop.run(500)
            torch.cuda.synchronize()
            tf = time.perf_counter()
            for _ in range(5):
                loop.new_scene(pts, poses_2d=p2d)
                loop.run(500)
            torch.cuda.synchronize()
            extras["frame_stream_ms"] = 1e3 * (time.perf_counter() - tf) / 5
        except Exception as e:
            extras["loop_error"] = repr(e)[:200]

        try:
            if args.no_dropin:
                raise StopIteration
            # the reference's own iteration (train.py:130-161) with the modules swapped and nothing else changed:
            # render() of ONE view through the drop-in gaussian_renderer, masked-L2 in tensor ops, autograd backward
            import types
            from gaussian_renderer import render_functions
            from skelsplat_amd.loop import l2_loss_gaussian
            render = render_functions["diff-gaussian-rasterization-" + wl["dataset"]]
            pipe = types.SimpleNamespace(debug=False, antialiasing=False, compute_cov3D_python=False, convert_SHs_python=False)
            bgc = torch.zeros(3, device=dev)
            gm2 = GaussianModel().create_from_points(ref_scene.pose_3d_init, scene.spatial_lr_scale, scene.n_joints,
                                                     scene_type=wl["dataset"], device=dev)
            gm2.training_setup()

            from skelsplat_amd.ops import l2_loss_gaussian as l2_loss_gaussian_fused

            def dropin_iteration(i, criterion):
                cam = scene.cameras[i % V]
                pkg = render(cam, gm2, pipe, bgc)
                loss, _ = criterion(pkg["render"], hm[i % V])   # (loss, error image) like loss_utils.py:100
                loss.backward()
                if (i + 1) % V == 0:
                    gm2.optimizer.step()
                    gm2.optimizer.zero_grad(set_to_none=True)

            # tensor-op criterion as in the reference, then the fused criterion registered in its `losses` table
            for tag, crit in (("dropin_iteration_ms", l2_loss_gaussian), ("dropin_iteration_fused_loss_ms", l2_loss_gaussian_fused)):
                for i in range(2 * V):
                    dropin_iteration(i, crit)
                torch.cuda.synchronize()
                td = time.perf_counter()
                nd = max(2 * V, args.steps // 4)
                for i in range(nd):
                    dropin_iteration(i, crit)
                torch.cuda.synchronize()
                extras[tag] = 1e3 * (time.perf_counter() - td) / nd
        except StopIteration:
            pass
        except Exception as e:
            extras["dropin_error"] = repr(e)[:200]

    if rank == 0:
        views_total = V * world * args.steps
        res = {
            "metric": "rendered+backpropagated views/s (differentiable skeletal-Gaussian rasterizer fwd+bwd)",
            "value": views_total / dt, "unit": "views/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": wl["name"], "views_per_gpu_step": V, "P": P, "C": C, "W": W, "H": H,
                       "parallelism": f"view-sharded x{world}" if world > 1 else "single GPU",
                       "path": "C ABI sks_forward + sks_backward, eager launches"},
        }
        if prof and fwd_n:
            alg_bytes = 4.0 * H * W * (C + 1) * V   # per launch: V views of (C colour + 1 inverse-depth) fp32 planes
            avg_s = fwd_ms * 1e-3 / fwd_n
            traffic = None
            tpath = os.path.join(ROOT, "profiles", "traffic.json")
            if os.path.exists(tpath):
                try:
                    traffic = json.load(open(tpath)).get(wl["name"], {}).get("fwd_bytes_per_launch")
                except Exception:
                    traffic = None
            res["roofline"] = {"bound": "hbm", "kernel": "k_render_fwd_sparse (forward fill + sparse compositor)",
                               "achieved": alg_bytes / avg_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": alg_bytes / avg_s / 1e9 / HBM_PEAK_GBS, "traffic": traffic,
                               "avg_launch_us": avg_s * 1e6, "launch_us_p10_p50_p90": [round(1e3 * x, 2) for x in fwd_q],
                               "launches_timed": fwd_n, "launches": args.steps,
                               "algorithmic_bytes_per_launch": alg_bytes}
            if bwd_n:
                res["bwd_kernel_avg_us"] = bwd_ms * 1e3 / bwd_n
                res["bwd_kernel_us_p10_p50_p90"] = [round(1e3 * x, 2) for x in bwd_q]
        res.update(extras)
        if not args.no_cpu_baseline and world == 1:
            res["cpu_baseline"] = cpu_baseline(ref_scene, params, n_views=2)
        os.write(real_stdout, (json.dumps(res) + "\n").encode())
    # nothing may follow the result line on either stream: what native libraries still hold in their stdio buffers (RCCL's
    # banner) or print while tearing down goes to /dev/null from here on
    sys.stdout.flush()
    sys.stderr.flush()
    devnull = os.open(os.devnull, os.O_WRONLY)
    os.dup2(devnull, 1)
    os.dup2(devnull, 2)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
