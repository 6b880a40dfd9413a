"""One fused-SSIM configuration in a loop (for rocprofv3 counter passes): python tools/ssim_one.py fwd|train|bwd [iters] [B,CH,H,W]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from skelsplat_amd import ops

mode = sys.argv[1] if len(sys.argv) > 1 else "fwd"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device("cuda:0")
shape = tuple(int(v) for v in sys.argv[3].split(",")) if len(sys.argv) > 3 else (5, 1, 1500, 1500)
a = torch.rand(shape, device=dev, requires_grad=mode != "fwd")
b = torch.rand(shape, device=dev)
for _ in range(iters):
    if mode == "fwd":
        with torch.no_grad():
            ops.FusedSSIMMap.apply(1e-4, 9e-4, a, b, "same", False)
    else:
        a.grad = None
        ops.fused_ssim(a, b).backward()
torch.cuda.synchronize()
