"""skelsplat_amd -- MI355X-native (gfx950) hot path of SkelSplat: differentiable skeletal-Gaussian rasterizer,
its render surface and the multi-view optimisation loop.  All compute lives in libskelsplat_hip.so
(hand-written HIP, C ABI in include/skelsplat_hip.h); PyTorch-ROCm is used for device memory, streams,
autograd plumbing and torch.distributed (RCCL)."""
from .rasterizer import (GaussianRasterizationSettings, GaussianRasterizer, ViewBatch, forward_views,  # noqa: F401
                         backward_views, rasterize_views, rasterize_gaussians, make_package)

__version__ = "0.1.0"
