#!/usr/bin/env python3
"""Forward of the real scenes by store kind (non-temporal / plain) x passes per fill block: kernel time (hipEvent hooks) and the
two-call step.  Plain stores let the 256 MB Infinity Cache hold back part of a call's writes when the SAME output buffers are written
step after step (a loop's Workspace) -- it pays when a call's output is about that size (H36M: 288 MB), not when it is many times
larger (Panoptic 5.1 GB, stress 2.4 GB).   python tools/sweep_nt.py [h36m|panoptic] [views]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from skelsplat_amd import _lib, rasterizer as R
from tools.tune_fwd import setup
import time


def run(views, params, dL, tune, iters=30, ws=None):
    """tools/tune_fwd.run on a Workspace: the same output buffers every call, as bench.py's step and the loops have them"""
    for _ in range(3):
        st = R.forward_views(views, *params, tune_flags=tune, workspace=ws)[3]
        R.backward_views(st, *params, dL, workspace=ws)
    torch.cuda.synchronize()
    _lib.prof_enable(True); _lib.prof_read(0); _lib.prof_read(1)
    t0 = time.perf_counter()
    for _ in range(iters):
        st = R.forward_views(views, *params, tune_flags=tune, workspace=ws)[3]
        R.backward_views(st, *params, dL, workspace=ws)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    f, fn = _lib.prof_read(0); b, bn = _lib.prof_read(1)
    _lib.prof_enable(False)
    return f / fn * 1e3, b / bn * 1e3, dt * 1e6


ds = sys.argv[1] if len(sys.argv) > 1 else "h36m"
V = int(sys.argv[2]) if len(sys.argv) > 2 else 4
scene, views, params, dL = setup(ds, V)
ws = R.Workspace()
for rep in range(2):
    for nt_off in (0, 16):
        row = []
        for pb in (0, 2, 3, 4, 5, 6, 8):
            f, b, tot = run(views, params, dL, (pb << 8) | nt_off, iters=60 if V <= 8 else 12, ws=ws)
            row.append(f"pb={pb or 'dflt'}: {f:.1f}/{tot:.1f}")
        print(ds, V, "plain" if nt_off else "nt   ", "fwd/step us:", ", ".join(row), flush=True)
