"""Drop-in for the reference's gaussian_renderer package (gaussian_renderer/__init__.py:28-371): the three
render_* functions and the `render_functions` registry keyed by `pipeline.rendering` (configs/*.yaml:45).

Differences that do not change results (fp32 round-off aside):
  * when the model is the reference's GaussianModel -- get_opacity / get_scaling / get_rotation are torch.sigmoid / torch.exp /
    F.normalize of the leaves (scene/gaussian_model.py:39-47,100-110) -- the leaves themselves go to the rasterizer and the
    three activations and their autograd run inside its kernels (`raw_params`): eight tiny launches less per render(), ten
    less per backward, in an iteration that is bound by its launches.  FAST_ACTIVATIONS = False (or SKS_RENDER_FAST=0)
    keeps the reference's literal call: the activated tensors, autograd through torch's activation nodes;
  * `rendered_image.clamp(0, 1)` (reference :129) is folded into the rasterizer's store and its backward mask
    (SKS_CLAMP01) instead of running as two extra dense passes;
  * `(radii > 0).nonzero()` (reference :133) forces a host sync, so `visibility_filter` is computed on first access.
"""
import math
import os

import torch

from diff_gaussian_rasterization_h36m import GaussianRasterizationSettings as GaussianRasterizationSettingsH36M
from diff_gaussian_rasterization_h36m import GaussianRasterizer as GaussianRasterizerH36M
from diff_gaussian_rasterization_panoptic import GaussianRasterizationSettings as GaussianRasterizationSettingsPanoptic
from diff_gaussian_rasterization_panoptic import GaussianRasterizer as GaussianRasterizerPanoptic
from diff_gaussian_rasterization_op import GaussianRasterizationSettings as GaussianRasterizationSettingsOp
from diff_gaussian_rasterization_op import GaussianRasterizer as GaussianRasterizerOp


FAST_ACTIVATIONS = os.environ.get("SKS_RENDER_FAST", "1") != "0"


_STOCK = {}     # model class -> are its three getters the reference's?


def _stock_getters(cls):
    """True when cls.get_opacity / get_scaling / get_rotation are properties that do what the reference's do and nothing else --
    `return self.<x>_activation(self._<x>)` (scene/gaussian_model.py:102-108,128-130): judged from the getter's code object (the
    two names it touches, no constants, no closure), so a subclass that filters its scales or masks its opacity in the getter, but
    keeps the stock activation attributes, takes the literal path and its getters (and their autograd) run."""
    hit = _STOCK.get(cls)
    if hit is None:
        def stock(name, act, leaf):
            p = getattr(cls, name, None)
            f = getattr(p, "fget", None) if isinstance(p, property) else None
            co = getattr(f, "__code__", None)
            return (co is not None and co.co_argcount == 1 and co.co_names == (act, leaf) and not co.co_freevars
                    and not f.__closure__ and all(k is None or isinstance(k, str) for k in co.co_consts))
        hit = _STOCK[cls] = (stock("get_opacity", "opacity_activation", "_opacity") and stock("get_scaling", "scaling_activation", "_scaling")
                             and stock("get_rotation", "rotation_activation", "_rotation"))
    return hit


def _leaves_of(pc):
    """(_opacity, _scaling, _rotation) if pc's three activations AND the getters through them are exactly the reference's, else
    None (the literal call: activated tensors, autograd through the model's own getters)."""
    try:
        if (pc.opacity_activation is torch.sigmoid and pc.scaling_activation is torch.exp
                and pc.rotation_activation is torch.nn.functional.normalize and _stock_getters(type(pc))):
            return pc._opacity, pc._scaling, pc._rotation
    except AttributeError:
        pass
    return None


class RenderPackage(dict):
    """dict with the reference's keys (gaussian_renderer/__init__.py:131-138).  `visibility_filter` =
    (radii > 0).nonzero() forces a host sync, so it is materialised on first use -- through ANY accessor: [], get, in,
    keys / values / items, iteration, len, copy, dict(pkg), ** unpacking all see the five keys of the reference's dict."""
    _LAZY = "visibility_filter"

    def _materialise(self):
        if not dict.__contains__(self, self._LAZY):
            dict.__setitem__(self, self._LAZY, (dict.__getitem__(self, "radii") > 0).nonzero())

    def __getitem__(self, key):
        if key == self._LAZY:
            self._materialise()
        return dict.__getitem__(self, key)

    def get(self, key, default=None):
        if key == self._LAZY:
            self._materialise()
        return dict.get(self, key, default)

    def __contains__(self, key):
        return key == self._LAZY or dict.__contains__(self, key)

    def __iter__(self):
        self._materialise()
        return dict.__iter__(self)

    def __len__(self):
        return dict.__len__(self) + (0 if dict.__contains__(self, self._LAZY) else 1)

    def keys(self):
        self._materialise()
        return dict.keys(self)

    def values(self):
        self._materialise()
        return dict.values(self)

    def items(self):
        self._materialise()
        return dict.items(self)

    def copy(self):
        self._materialise()
        return dict(dict.items(self))

    def __repr__(self):
        self._materialise()
        return dict.__repr__(self)


def _render(Settings, Rasterizer, viewpoint_camera, pc, pipe, bg_color, scaling_modifier=1.0, separate_sh=False,
            override_color=None, use_trained_exp=False):
    # zero tensor whose gradient receives the 2D (screen-space) mean gradients (reference :36-40)
    # (the reference adds 0 to make it a non-leaf and calls retain_grad(); a leaf keeps its .grad by itself: one launch less)
    screenspace_points = torch.zeros_like(pc.get_xyz, dtype=pc.get_xyz.dtype, requires_grad=True)
    tanfovx = math.tan(viewpoint_camera.FoVx * 0.5)
    tanfovy = math.tan(viewpoint_camera.FoVy * 0.5)
    raster_settings = Settings(
        image_height=int(viewpoint_camera.image_height),
        image_width=int(viewpoint_camera.image_width),
        tanfovx=tanfovx,
        tanfovy=tanfovy,
        bg=bg_color,
        scale_modifier=scaling_modifier,
        viewmatrix=viewpoint_camera.world_view_transform,
        projmatrix=viewpoint_camera.full_proj_transform,
        sh_degree=pc.active_sh_degree,
        campos=viewpoint_camera.camera_center,
        prefiltered=False,
        debug=pipe.debug,
        antialiasing=pipe.antialiasing,
    )
    rasterizer = Rasterizer(raster_settings=raster_settings)
    means3D = pc.get_xyz
    means2D = screenspace_points
    leaves = _leaves_of(pc) if (FAST_ACTIVATIONS and not pipe.compute_cov3D_python) else None
    scales = rotations = cov3D_precomp = None
    if leaves is not None:
        opacity, scales, rotations = leaves
    elif pipe.compute_cov3D_python:
        opacity = pc.get_opacity
        cov3D_precomp = pc.get_covariance(scaling_modifier)
    else:
        opacity = pc.get_opacity
        scales = pc.get_scaling
        rotations = pc.get_rotation
    shs = colors_precomp = None
    if override_color is None:
        if pipe.convert_SHs_python:
            # Refused, not stubbed (INTEGRATION.md "Switches that are refused"): the reference's branch
            # (gaussian_renderer/__init__.py:85-90) views the (J,1,J) features as (-1, 3, (deg+1)^2) RGB SH -- a shape error for
            # J = 17 / 19 (J*J is not a multiple of 3) and a (75,3) colour table for J = 15 -- and then calls the rasterizer
            # with shs=None, whose kernels read the features from the `sh` pointer (quirk Q1): a null dereference.
            raise RuntimeError("pipe.convert_SHs_python=True cannot work with skeleton features (P,1,J): the reference's "
                               "own branch fails on them (see INTEGRATION.md); every shipped config sets it to false "
                               "(configs/h36m.yaml:46)")
        shs = pc.get_features  # with separate_sh the reference passes dc + empty rest; same (P,1,C) features
    else:
        colors_precomp = override_color
    rendered_image, radii, depth_image = rasterizer(
        means3D=means3D, means2D=means2D, shs=shs, colors_precomp=colors_precomp, opacities=opacity,
        scales=scales, rotations=rotations, cov3D_precomp=cov3D_precomp, clamp01=True, raw_params=leaves is not None)
    return RenderPackage(render=rendered_image, viewspace_points=screenspace_points, radii=radii, depth=depth_image)


def render_h36m(viewpoint_camera, pc, pipe, bg_color, scaling_modifier=1.0, separate_sh=False, override_color=None,
                use_trained_exp=False):
    return _render(GaussianRasterizationSettingsH36M, GaussianRasterizerH36M, viewpoint_camera, pc, pipe, bg_color,
                   scaling_modifier, separate_sh, override_color, use_trained_exp)


def render_panoptic(viewpoint_camera, pc, pipe, bg_color, scaling_modifier=1.0, separate_sh=False, override_color=None,
                    use_trained_exp=False):
    return _render(GaussianRasterizationSettingsPanoptic, GaussianRasterizerPanoptic, viewpoint_camera, pc, pipe,
                   bg_color, scaling_modifier, separate_sh, override_color, use_trained_exp)


def render_op(viewpoint_camera, pc, pipe, bg_color, scaling_modifier=1.0, separate_sh=False, override_color=None,
              use_trained_exp=False):
    return _render(GaussianRasterizationSettingsOp, GaussianRasterizerOp, viewpoint_camera, pc, pipe, bg_color,
                   scaling_modifier, separate_sh, override_color, use_trained_exp)


render_functions = {
    "diff-gaussian-rasterization-h36m": render_h36m,
    "diff-gaussian-rasterization-panoptic": render_panoptic,
    "diff-gaussian-rasterization-op": render_op,
}
