"""Seeded rasterizer test cases shared by the CPU (oracle) and GPU (parity) tests."""
import math

import numpy as np

from oracle import oracle as orc
from skelsplat_amd.scene import SyntheticScene


class Case:
    pass


def make_case(seed, W, H, dataset="h36m", n_views=2, scale_log=4.0, rand_rot=True, opac=None, fxmul=1.0,
              ring=2500.0, n_skeletons=1, onehot=False, pitch=700.0, with_dL=True):
    """Skeleton(s) seen by `n_views` ring cameras; anisotropic random covariances so splats overlap and saturate."""
    sc = SyntheticScene(dataset, n_views=n_views, seed=seed, W=W, H=H, ring=ring,
                        fx=1145.0 * (W / 1000) * fxmul, n_skeletons=n_skeletons, pitch=pitch)
    from skelsplat_amd.scene import random_gaussian_params
    g = random_gaussian_params(sc, seed, scale_log=scale_log, rand_rot=rand_rot, opac=opac, onehot=onehot)
    rng = g["_rng"]
    c = Case()
    c.scene = sc
    c.W, c.H, c.P, c.C = W, H, sc.n_points, sc.n_joints
    c.means, c.scales, c.quats, c.opac, c.feat = g["means"], g["scales"], g["quats"], g["opac"], g["feat"]
    c.cams = sc.cameras
    c.ocams = [orc.Cam(W, H, math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5),
                       cam.world_view_transform.numpy(), cam.full_proj_transform.numpy()) for cam in c.cams]
    if with_dL:    # (drawn last: leaving them out -- full-size cases make theirs on the GPU -- changes nothing above)
        c.dL_color = rng.normal(0, 1, (n_views, c.C, H, W)).astype(np.float32)
        c.dL_inv = rng.normal(0, 1, (n_views, 1, H, W)).astype(np.float32)
    return c


def oracle_forward(c, v, antialiasing=False):
    return orc.forward(c.means, c.feat, c.opac, c.scales, c.quats, None, c.ocams[v], antialiasing=antialiasing)


def oracle_backward(c, v, fwd, antialiasing=False, bg=None, with_inv=True):
    return orc.backward(fwd, c.means, c.feat, c.opac, c.scales, c.quats, None, c.ocams[v], c.dL_color[v],
                        c.dL_inv[v] if with_inv else None, bg=bg, antialiasing=antialiasing)


def assert_close(name, got, want, rtol=1e-3, atol_scale=1e-5):
    got = np.asarray(got, dtype=np.float64)
    want = np.asarray(want, dtype=np.float64)
    assert got.shape == want.shape, (name, got.shape, want.shape)
    scale = np.abs(want).max() + 1e-30
    err = np.abs(got - want)
    tol = atol_scale * scale + rtol * np.abs(want)
    bad = err > tol
    assert not bad.any(), f"{name}: {bad.sum()} / {bad.size} off; max abs err {err.max():.3e} (scale {scale:.3e})"


EPS32 = 2.0 ** -24
# How far two fp32 evaluations of one gradient may sit apart, in units of 2**-24 x (the oracle's sum of |terms| over every pixel
# and every path of the geometry backward, oracle.backward(bounds=True)).  Calibrated with tools/fuzz_bound_calib.py on the
# MI355X: the largest excess seen over 4 000 `extreme` cases x 7 gradients is 2.9 units (profiles/r05_fuzz_bound_calib.txt; below one unit for five of the seven);
# the constant leaves a factor ~5 over it.  A dropped 16x16 tile of a 100-tile footprint moves a sum by ~1e-2 of its |terms|:
# four orders of magnitude above this allowance.
BOUND_KAPPA = 16.0


def bound_excess(got, want, bound, rtol=1e-3):
    """max over elements of (|got - want| - rtol |want|) / (2**-24 x bound): the units BOUND_KAPPA is stated in."""
    got, want, bound = (np.asarray(a, dtype=np.float64) for a in (got, want, bound))
    ex = np.abs(got - want) - rtol * np.abs(want)
    with np.errstate(divide="ignore", invalid="ignore"):
        r = np.where(ex > 0, ex / (EPS32 * bound), 0.0)
    return float(np.nan_to_num(r, nan=np.inf, posinf=np.inf).max()) if r.size else 0.0


def assert_close_bound(name, got, want, bound, rtol=1e-3, kappa=None):
    """|got - want| <= rtol |want| + kappa 2**-24 bound, elementwise: a relative tolerance plus a COMPUTED rounding allowance
    (no blanket fraction of the tensor's largest entry)."""
    kappa = BOUND_KAPPA if kappa is None else kappa
    got = np.asarray(got, dtype=np.float64)
    want = np.asarray(want, dtype=np.float64)
    bound = np.asarray(bound, dtype=np.float64)
    assert got.shape == want.shape == bound.shape, (name, got.shape, want.shape, bound.shape)
    err = np.abs(got - want)
    tol = rtol * np.abs(want) + kappa * EPS32 * bound
    bad = ~(err <= tol)
    assert not bad.any(), (f"{name}: {bad.sum()} / {bad.size} off; max abs err {err.max():.3e}, "
                           f"worst excess {bound_excess(got, want, bound, rtol):.1f} x 2^-24 x sum|terms| (allowed {kappa:g})")
