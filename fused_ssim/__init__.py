"""Drop-in for the reference package `fused_ssim` (submodules/fused-ssim/fused_ssim/__init__.py), backed by the
MI355X HIP library of skelsplat_amd."""
from skelsplat_amd.ops import FusedSSIMMap, fused_ssim, allowed_padding  # noqa: F401
