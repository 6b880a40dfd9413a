"""In-situ floor: how long does a plain fill of the same (4,18,1000,1000) fp32 output take when it alternates between
two buffers and interleaves with the backward kernels, exactly like the benchmark loop?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from skelsplat_amd import rasterizer as R
from tools.tune_fwd import setup

scene, views, params, dL = setup("h36m", 4)
bufs = [torch.empty((4, 18, 1000, 1000), device="cuda") for _ in range(2)]
c, i, r, st = R.forward_views(views, *params)
for name, fn in (("tensor.zero_()", lambda b: b.zero_()), ("tensor.fill_(1)", lambda b: b.fill_(1.0))):
    ts = []
    for it in range(40):
        b = bufs[it % 2]
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(b); e1.record()
        R.backward_views(st, *params, dL)
        torch.cuda.synchronize()
        if it >= 5:
            ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    print(f"{name}: median {ts[len(ts)//2]:.1f} us, min {ts[0]:.1f} us  ({288e6/ts[len(ts)//2]/1e3:.0f} GB/s)")

from skelsplat_amd import _lib
# the forward and backward kernels in the same interleaving (hipEvent pairs around the kernel, like bench.py).
# Experiments that switched pieces of the forward off (no cover look-up / no composite blocks / passes per block /
# composite slots) were run from this script with temporary kernel flags; their numbers are in DESIGN.md section 5.
for name, tune in (("sks fwd (row-aligned fill)", 0), ("sks fwd, linear fill", 1 << 21), ("sks fwd, row-aligned, 1 row per block", 1 << 8), ("sks fwd, row-aligned, 4 rows per block", 4 << 8)):
    _lib.prof_enable(True); _lib.prof_read(0)
    for it in range(40):
        c, i, r, st2 = R.forward_views(views, *params, tune_flags=tune)
        R.backward_views(st, *params, dL)
        keep = (c, i)
    torch.cuda.synchronize()
    ms, n = _lib.prof_read(0); _lib.prof_enable(False)
    print(f"{name}: avg {ms/n*1e3:.1f} us")

_lib.prof_enable(True); _lib.prof_read(1)
for it in range(40):
    c, i, r, st2 = R.forward_views(views, *params)
    R.backward_views(st2, *params, dL)
torch.cuda.synchronize()
ms, n = _lib.prof_read(1); _lib.prof_enable(False)
print(f"sks bwd: avg {ms/n*1e3:.1f} us")
