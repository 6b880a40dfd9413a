"""`simple_knn._C` of the reference is a pybind module exposing distCUDA2 (submodules/simple-knn/ext.cpp)."""
from skelsplat_amd.ops import distCUDA2  # noqa: F401
