"""Duration of the forward kernel INSIDE the one-call step (the backward beside it on the second queue), by the library's own
hipEvent pairs, and the step time -- across SKS_BWD_WG (workgroups per (view, Gaussian) of the backward) and passes per fill block."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from skelsplat_amd import rasterizer as R, _lib

dev = torch.device("cuda:0")
wl = bench.WORKLOADS["h36m"]
scene, gm, params = bench.make_scene(torch, wl, dev)
views = R.ViewBatch.from_cameras(scene.cameras)
dL = torch.randn((4, 17, 1000, 1000), device=dev)
for one_call in (True, False):
    for tune in (0, 3):
        step = bench.ApiStep(views, params, dL, one_call=one_call)
        step(); step()
        step.ws._plans["fwd"][2][16] |= tune << 8
        for _ in range(20):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(300):
            step()
        torch.cuda.synchronize()
        us = 1e6 * (time.perf_counter() - t0) / 300
        _lib.prof_enable(True, every=4, kinds=(0,))
        _lib.prof_read(0)
        for _ in range(200):
            step()
        torch.cuda.synchronize()
        tot, n, q = _lib.prof_read_quantiles(0)
        _lib.prof_enable(False)
        print(f"one_call={one_call} passes={tune or 2} SKS_BWD_WG={os.environ.get('SKS_BWD_WG', '0')}: step {us:.1f} us, forward kernel mean {1e3 * tot / n:.1f} us "
              f"(p10/p50/p90 {1e3 * q[0]:.1f}/{1e3 * q[1]:.1f}/{1e3 * q[2]:.1f}) = {288e6 / (tot / n * 1e-3) / 8e12:.3f} of peak", flush=True)
