"""Fused-SSIM timings (hipEvent-free: wall clock over back-to-back launches) at the shapes of BASELINE.md §1/§3."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from skelsplat_amd import ops

dev = torch.device("cuda:0")


def timeit(fn, iters=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e6


for shape in ((5, 1, 1500, 1500), (5, 5, 1080, 1920), (1, 17, 1000, 1000), (4, 17, 1000, 1000)):
    a = torch.rand(shape, device=dev, requires_grad=True)
    b = torch.rand(shape, device=dev)
    npx = a.numel()

    def train_mean():
        a.grad = None
        ops.fused_ssim(a, b).backward()

    def train_map():
        a.grad = None
        ops.FusedSSIMMap.apply(1e-4, 9e-4, a, b, "same", True).mean().backward()

    with torch.no_grad():
        t_inf = timeit(lambda: ops.fused_ssim(a, b, train=False))
        t_map = timeit(lambda: ops.FusedSSIMMap.apply(1e-4, 9e-4, a, b, "same", False))
    t_fwd = timeit(lambda: ops.fused_ssim(a, b))
    t_tr = timeit(train_mean)
    t_trm = timeit(train_map)
    print(f"{shape}: mean inference {t_inf:.0f} us ({8 * npx / t_inf / 1e6:.2f} TB/s of 2 reads) | map inference {t_map:.0f} us "
          f"({12 * npx / t_map / 1e6:.2f} TB/s) | train fwd {t_fwd:.0f} us ({20 * npx / t_fwd / 1e6:.2f} TB/s) | "
          f"train fwd+bwd {t_tr:.0f} us ({44 * npx / t_tr / 1e6:.2f} TB/s of 44 B/px) | via map API {t_trm:.0f} us", flush=True)
