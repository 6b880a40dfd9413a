"""What does tensor.zero_() launch for the forward's 288 MB?  (rocprofv3 --kernel-trace -- python3 tools/zero_shape.py)"""
import torch
b = torch.empty((4, 18, 1000, 1000), device="cuda")
for _ in range(20):
    b.zero_()
torch.cuda.synchronize()
