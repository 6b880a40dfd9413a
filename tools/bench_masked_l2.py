"""k_masked_l2 (read render + gt, write dL: 3 dense passes) against a torch add of the same tensors (the same 3 passes) and a copy."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from skelsplat_amd.ops import masked_l2

dev = torch.device("cuda:0")


def timed(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return 1e6 * (time.perf_counter() - t0) / n


for shape in ((1, 17, 1000, 1000), (4, 17, 1000, 1000), (31, 19, 1080, 1920)):
    r = torch.rand(shape, device=dev) * (torch.rand(shape, device=dev) > 0.9)
    g = torch.rand(shape, device=dev) * (torch.rand(shape, device=dev) > 0.9)
    out = torch.empty_like(r)
    gb = 3 * r.numel() * 4 / 1e9
    t_l2 = timed(lambda: masked_l2(r, g))
    t_add = timed(lambda: torch.add(r, g, out=out))
    t_l2_nograd = timed(lambda: masked_l2(r, g, want_grad=False))
    print(f"{shape}: masked_l2 {t_l2:.1f} us = {gb / t_l2 * 1e3:.2f} TB/s | torch.add {t_add:.1f} us = {gb / t_add * 1e3:.2f} TB/s | "
          f"masked_l2 without dL {t_l2_nograd:.1f} us = {gb * 2 / 3 / t_l2_nograd * 1e3:.2f} TB/s")
