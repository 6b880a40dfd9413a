#!/usr/bin/env python3
"""Host vs GPU cost of the exchange step at world 1 (RCCL): issue time of all_gather_into_tensor, and the GPU gap around it."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, torch.distributed as dist
from skelsplat_amd import rasterizer as R
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29655")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
P = 19
shard = torch.zeros((4, P, 3), device=dev); allg = torch.zeros((32, P, 3), device=dev); out = torch.empty((P, 3), device=dev)
big = torch.empty(64 << 20, device=dev)
def step(gather=True):
    big.zero_()                                   # ~40 us of GPU work in front, so the host runs ahead
    if gather:
        dist.all_gather_into_tensor(allg[:4], shard)
    R.mean_views(allg, 31, 8, out=out)
for g in (True, False):
    for _ in range(20): step(g)
    torch.cuda.synchronize()
    n = 300
    t0 = time.perf_counter()
    for _ in range(n): step(g)
    ti = time.perf_counter() - t0
    torch.cuda.synchronize()
    ta = time.perf_counter() - t0
    print(f"gather={g}: host issue {1e6*ti/n:.1f} us/step, complete {1e6*ta/n:.1f} us/step")
t0 = time.perf_counter()
for _ in range(300): dist.all_gather_into_tensor(allg[:4], shard)
ti = time.perf_counter() - t0
torch.cuda.synchronize()
print(f"all_gather alone: host issue {1e6*ti/300:.1f} us, complete {1e6*(time.perf_counter()-t0)/300:.1f} us")
from skelsplat_amd.rccl_direct import DirectGather
dg = DirectGather.create(dev)
print("direct communicator:", dg is not None)
if dg is not None:
    def step_direct():
        big.zero_()
        dg.all_gather_into_tensor(allg[:4], shard)
        R.mean_views(allg, 31, 8, out=out)
    shard.normal_()
    for _ in range(20): step_direct()
    torch.cuda.synchronize()
    assert torch.equal(allg[:4], shard)
    t0 = time.perf_counter()
    for _ in range(300): step_direct()
    ti = time.perf_counter() - t0
    torch.cuda.synchronize()
    ta = time.perf_counter() - t0
    print(f"direct ncclAllGather on the current stream: host issue {1e6*ti/300:.1f} us/step, complete {1e6*ta/300:.1f} us/step")
    dg.destroy()
dist.destroy_process_group()
