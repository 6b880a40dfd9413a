"""Runs the REFERENCE's own Python (/root/reference) in the build container, for the fixture generators only.

TEST INFRASTRUCTURE, BUILD CONTAINER ONLY: nothing here is imported by the tests or the product (the reference does not
exist on the GPU box); the generators that use it commit small .npz fixtures.

What stands in for what, so that the unmodified reference modules import and run without a GPU, CUDA or the absent
third-party packages:
  * `_C` of the three diff_gaussian_rasterization_* packages (the pybind module built from the CUDA sources, DGR/ext.cpp:14-18)
    -> `OracleC`: same three entry points, same positional signatures and return tuples (DGR/rasterize_points.h:18-71),
    computed by oracle/sks_oracle.c; every call is recorded.  The reference's Python around it -- GaussianRasterizer,
    _RasterizeGaussians (20-argument forward, 24-argument backward, 9-slot gradient return), render_h36m / _panoptic / _op,
    train.training() -- is the reference's own code, imported from where it lies.
  * device strings: a TorchFunctionMode maps device="cuda" / .cuda() / .to("cuda") to the CPU; torch.cuda.Event /
    synchronize / empty_cache are no-ops.
  * absent packages: tensordict.TensorDict -> dict, cupy.asarray -> numpy, cupyx.scipy.ndimage.gaussian_filter ->
    scipy.ndimage.gaussian_filter (cupy mirrors scipy's API and defaults), plyfile / cv2 / hydra / omegaconf / open3d ->
    empty modules (nothing on the executed path calls them), fused_ssim / diff_gaussian_rasterization -> ImportError
    (train.py:41-51 guards both imports).
"""
import contextlib
import importlib.util
import os
import sys
import types

import numpy as np
import torch
from torch.overrides import TorchFunctionMode

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PACKAGES = {"h36m": 17, "panoptic": 19, "op": 15}       # NUM_CHANNELS, DGR*/cuda_rasterizer/config.h:15


def _is_cuda(d):
    if isinstance(d, str):
        return d.startswith("cuda")
    return isinstance(d, torch.device) and d.type == "cuda"


class CudaToCpu(TorchFunctionMode):
    def __torch_function__(self, func, types_, args=(), kwargs=None):
        kwargs = dict(kwargs or {})
        if _is_cuda(kwargs.get("device")):
            kwargs["device"] = "cpu"
        name = getattr(func, "__name__", "")
        if name == "cuda" and args and isinstance(args[0], torch.Tensor):
            return args[0]
        if name == "to":
            args = tuple("cpu" if _is_cuda(a) else a for a in args)
        return func(*args, **kwargs)


class _Event:
    def __init__(self, *a, **k):
        pass

    def record(self, *a, **k):
        pass

    def elapsed_time(self, other):
        return 0.0


@contextlib.contextmanager
def no_gpu():
    saved = (torch.cuda.Event, torch.cuda.synchronize, torch.cuda.empty_cache)
    torch.cuda.Event, torch.cuda.synchronize, torch.cuda.empty_cache = _Event, (lambda *a, **k: None), (lambda: None)
    try:
        with CudaToCpu():
            yield
    finally:
        torch.cuda.Event, torch.cuda.synchronize, torch.cuda.empty_cache = saved


class OracleC:
    """Stand-in for the compiled `_C` of one rasterizer package; `calls` records every invocation."""

    def __init__(self, num_channels, calls):
        self.C, self.calls, self._fwd = num_channels, calls, {}
        if ROOT not in sys.path:
            sys.path.append(ROOT)
        from oracle import oracle as orc
        self.orc = orc

    @staticmethod
    def _np(t):
        return None if t is None or t.numel() == 0 else t.detach().cpu().numpy()

    @staticmethod
    def _snap(args):
        return tuple(a.detach().clone() if isinstance(a, torch.Tensor) else a for a in args)

    def rasterize_gaussians(self, *args):
        (bg, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D_precomp, viewmatrix, projmatrix, tan_fovx,
         tan_fovy, image_height, image_width, sh, degree, campos, prefiltered, antialiasing, debug) = args
        self.calls.append(("rasterize_gaussians", self._snap(args)))
        if means3D.dim() != 2 or means3D.shape[1] != 3:
            raise RuntimeError("means3D must have dimensions (num_points, 3)")          # rasterize_points.cu:58-60
        P = means3D.shape[0]
        assert sh.numel() and sh.shape[1] == 1 and sh.shape[2] == self.C, "features are read from `sh` as flat (P, NUM_CHANNELS)"
        cam = self.orc.Cam(image_width, image_height, tan_fovx, tan_fovy, self._np(viewmatrix), self._np(projmatrix))
        o = self.orc.forward(self._np(means3D), self._np(sh).reshape(P, self.C), self._np(opacity), self._np(scales),
                             self._np(rotations), self._np(cov3D_precomp), cam, scale_modifier, bool(antialiasing))
        key = len(self._fwd) + 1
        self._fwd[key] = (o, cam)
        geom = torch.tensor([key], dtype=torch.int64).view(torch.uint8)          # the handle travels in geomBuffer
        t = torch.from_numpy
        self.last_fwd_out = (int(o["R"]), o["color"].copy(), o["radii"].copy(), o["invdepth"].copy())
        return (int(o["R"]), t(o["color"]), t(o["radii"]), geom, torch.zeros(8, dtype=torch.uint8),
                torch.zeros(8, dtype=torch.uint8), t(o["invdepth"]))

    def rasterize_gaussians_backward(self, *args):
        (bg, means3D, radii, colors, opacities, scales, rotations, scale_modifier, cov3D_precomp, viewmatrix, projmatrix,
         tan_fovx, tan_fovy, dL_dout_color, dL_dout_invdepth, sh, degree, campos, geomBuffer, R, binningBuffer, imageBuffer,
         antialiasing, debug) = args
        self.calls.append(("rasterize_gaussians_backward", self._snap(args)))
        o, cam = self._fwd.pop(int(geomBuffer.view(torch.int64)[0]))
        P = means3D.shape[0]
        b = self.orc.backward(o, self._np(means3D), self._np(sh).reshape(P, self.C), self._np(opacities), self._np(scales),
                              self._np(rotations), self._np(cov3D_precomp), cam, self._np(dL_dout_color),
                              self._np(dL_dout_invdepth), bg=self._np(bg), scale_modifier=scale_modifier,
                              antialiasing=bool(antialiasing))
        t = lambda a, *s: torch.zeros(s) if a is None else torch.from_numpy(np.ascontiguousarray(a)).reshape(s)
        # rasterize_points.cu:222: (dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations).
        # dL_dsh: the reference's kernel leaves garbage there; the stand-in returns zeros.
        out = (t(b["dL_dmeans2D"], P, 3), t(b["dL_dcolors"], P, self.C), t(b["dL_dopacity"], P, 1), t(b["dL_dmeans3D"], P, 3),
               t(b["dL_dcov3D"], P, 6), torch.zeros((P, 1, self.C)), t(b["dL_dscales"], P, 3), t(b["dL_drotations"], P, 4))
        self.last_bwd_out = tuple(x.clone() for x in out)
        return out

    def mark_visible(self, means3D, viewmatrix, projmatrix):
        self.calls.append(("mark_visible", self._snap((means3D, viewmatrix, projmatrix))))
        cam = self.orc.Cam(16, 16, 1.0, 1.0, self._np(viewmatrix), self._np(projmatrix))
        return torch.from_numpy(self.orc.mark_visible(self._np(means3D), cam))


def _module(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def install(calls):
    """Stubs + the three rasterizer packages (the reference's __init__.py over an OracleC) + /root/reference on sys.path.
    Returns {package key: OracleC}."""
    import scipy.ndimage
    for n in ("tensordict", "cupy", "cupyx", "cupyx.scipy", "cupyx.scipy.ndimage", "plyfile", "cv2", "open3d", "hydra",
              "hydra.core", "hydra.core.hydra_config", "omegaconf"):
        _module(n)
    for n in ("hydra", "hydra.core", "cupyx", "cupyx.scipy"):
        sys.modules[n].__path__ = []
    sys.modules["hydra.core.hydra_config"].HydraConfig = object
    sys.modules["hydra"].main = lambda **k: (lambda f: f)
    sys.modules["omegaconf"].DictConfig = dict
    sys.modules["omegaconf"].OmegaConf = object
    sys.modules["tensordict"].TensorDict = dict
    sys.modules["cupy"].asarray = lambda t: np.asarray(t)
    sys.modules["cupyx.scipy.ndimage"].gaussian_filter = scipy.ndimage.gaussian_filter
    sys.modules["plyfile"].PlyData = sys.modules["plyfile"].PlyElement = None
    sys.modules["fused_ssim"] = None                        # -> ImportError, train.py:41-45
    sys.modules["diff_gaussian_rasterization"] = None       # -> ImportError, train.py:47-51, gaussian_model.py:25-28
    stubs = {}
    for key, C in PACKAGES.items():
        name = f"diff_gaussian_rasterization_{key}"
        pkg_dir = os.path.join(REF, "submodules", f"diff-gaussian-rasterization-{key}", name)
        stubs[key] = sys.modules[name + "._C"] = OracleC(C, calls)
        spec = importlib.util.spec_from_file_location(name, os.path.join(pkg_dir, "__init__.py"),
                                                      submodule_search_locations=[pkg_dir])
        mod = importlib.util.module_from_spec(spec)
        sys.modules[name] = mod
        spec.loader.exec_module(mod)
        assert mod._C is stubs[key]
    sys.path.insert(0, REF)
    if ROOT not in sys.path:
        sys.path.append(ROOT)       # behind the reference: `gaussian_renderer`, `scene`, `utils` resolve to /root/reference
    return stubs


class Cfg(dict):
    """A YAML mapping with attribute access (what train.py uses of omegaconf's DictConfig)."""
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__


def load_config(name, **overrides):
    """configs/<name>.yaml of the reference as nested Cfg; overrides: 'group.key' -> value."""
    import yaml

    def wrap(x):
        if isinstance(x, dict):
            return Cfg({k: wrap(v) for k, v in x.items()})
        if isinstance(x, str):          # PyYAML (YAML 1.1) reads `1e-5` as a string; OmegaConf, like YAML 1.2, as a float
            try:
                return float(x)
            except ValueError:
                return x
        return x

    cfg = wrap(yaml.safe_load(open(os.path.join(REF, "configs", name + ".yaml"))))
    for k, v in overrides.items():
        grp, key = k.split(".")
        cfg[grp][key] = v
    return cfg
