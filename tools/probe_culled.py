#!/usr/bin/env python3
"""The forward of an H36M-sized call with EVERY Gaussian culled (the launch is fill blocks only) next to `zero_()` of the same bytes,
per fill-block size: what the fill role costs by itself.   [SKS_LIB_OVERRIDE=variant.so] python tools/probe_culled.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from skelsplat_amd import _lib, rasterizer as R
from tools.tune_fwd import setup

scene, views, params, dL = setup("h36m", 4)
far = (params[0] + torch.tensor([0.0, 0.0, 1e9], device=params[0].device),) + params[1:]
NWS = int(os.environ.get("NWS", "1"))       # output buffers rotated over (1: the same 288 MB again and again, as a loop does; 8: 2.3 GB,
wss = [R.Workspace() for _ in range(NWS)]   # more than the 256 MB Infinity Cache can hold back)
xs = [torch.empty(288_000_000 // 4, device="cuda") for _ in range(NWS)]
for rep in range(3):
    row = []
    for nt_off in (0, 16):      # 16 = SKS_NO_NT_STORES
      for pb in (0, 1, 2, 3):
        tf = (pb << 8) | nt_off
        for i in range(3 * NWS):
            R.forward_views(views, *far, tune_flags=tf, workspace=wss[i % NWS])
        torch.cuda.synchronize()
        _lib.prof_enable(True); _lib.prof_read(0)
        for i in range(40):
            R.forward_views(views, *far, tune_flags=tf, workspace=wss[i % NWS])
        torch.cuda.synchronize()
        f, n = _lib.prof_read(0)
        _lib.prof_enable(False)
        row.append(f"{'plain' if nt_off else 'nt'} pb={pb or 'default'}: {f / n * 1e3:.1f}")
    ts = []
    s, e = torch.cuda.Event(True), torch.cuda.Event(True)
    for i in range(30):
        s.record(); xs[i % NWS].zero_(); e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e) * 1e3)
    ts.sort()
    print(os.path.basename(os.environ.get("SKS_LIB_OVERRIDE", "shipped")), f"NWS={NWS} culled forward us:", ", ".join(row), f"| zero_ p50 {ts[15]:.1f}", flush=True)
