"""Drop-in for the reference package `diff_gaussian_rasterization_op`
(submodules/diff-gaussian-rasterization-op, NUM_CHANNELS 15 at cuda_rasterizer/config.h:15), backed by the
MI355X HIP library of skelsplat_amd.  Same names as the reference's __init__.py:143-207."""
from skelsplat_amd.rasterizer import make_package, rasterize_gaussians  # noqa: F401

NUM_CHANNELS = 15
GaussianRasterizationSettings, GaussianRasterizer = make_package(NUM_CHANNELS)
