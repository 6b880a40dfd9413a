#!/usr/bin/env python3
"""The H36M forward kernel over ONE output set rewritten every call vs N output sets in turn (N x 288 MB: beyond the 256 MB Infinity
Cache), real scene vs every Gaussian culled (fill blocks only), non-temporal vs plain stores, next to `zero_()` of the same bytes
over the same rotation.  What bench.py's `roofline.frac` (8 sets in turn) and `frac_same_buffer` are made of."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from skelsplat_amd import _lib, rasterizer as R
from tools.tune_fwd import setup

scene, views, params, dL = setup("h36m", 4)
far = (params[0] + torch.tensor([0.0, 0.0, 1e9], device=params[0].device),) + params[1:]
NMAX = 8
wss = [R.Workspace() for _ in range(NMAX)]
xs = [torch.empty(288_000_000 // 4, device="cuda") for _ in range(NMAX)]


def fwd_us(p, nws, tf, with_bwd=False, n=48):
    def call(i):
        ws = wss[i % nws]
        st = R.forward_views(views, *p, tune_flags=tf, workspace=ws)[3]
        if with_bwd:
            R.backward_views(st, *p, dL, workspace=ws)
    for i in range(3 * nws):
        call(i)
    torch.cuda.synchronize()
    _lib.prof_enable(True, every=1, kinds=(0,)); _lib.prof_read(0)
    for i in range(n):
        call(i)
    torch.cuda.synchronize()
    tot, cnt, q = _lib.prof_read_quantiles(0)
    _lib.prof_enable(False)
    return q[1] * 1e3


def zero_us(nws):
    ts = []
    s, e = torch.cuda.Event(True), torch.cuda.Event(True)
    for i in range(40):
        s.record(); xs[i % nws].zero_(); e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e) * 1e3)
    ts.sort()
    return ts[20]


for rep in range(2):
    for nws in (1, 2, 4, 8):
        row = [f"sets={nws}"]
        for name, p in (("scene", params), ("culled", far)):
            for sname, bits in (("nt", 0), ("nt3", 3 << 8), ("plain3", (3 << 8) | 16)):
                row.append(f"{name}/{sname} {fwd_us(p, nws, bits):.1f}")
        row.append(f"scene/nt+bwd {fwd_us(params, nws, 0, True):.1f}")
        row.append(f"zero_ {zero_us(nws):.1f}")
        print("  ".join(row), flush=True)
