"""GPU parity: HIP rasterizer (through the C ABI) vs the CPU oracle on seeded cases.

Bars: integer artefacts (radii, tile rects, point_list, ranges, n_contrib) bit-exact; forward images
bit-exact as well (the compositor uses a fixed-op-sequence expf on both sides), asserted with a tiny fp32 tolerance
fallback only for documentation; gradients rtol 1e-3 / atol 1e-5*scale (fp32 atomics order).
"""
import numpy as np
import pytest
import torch

from tests import util
from skelsplat_amd import rasterizer as R

pytestmark = pytest.mark.gpu

CASES = [
    dict(seed=0, W=160, H=128, scale_log=4.0),
    dict(seed=1, W=200, H=120, scale_log=4.5, opac=1.0),
    dict(seed=2, W=96, H=96, scale_log=5.0, rand_rot=False, opac=1.0),
    dict(seed=3, W=130, H=77, scale_log=4.2),                      # W % 4 != 0: scalar store path, ragged tiles
    dict(seed=4, W=64, H=48, scale_log=3.0, dataset="panoptic"),   # C = 19
    dict(seed=5, W=176, H=144, scale_log=3.5, dataset="occlusion-person", n_skeletons=4),  # C = 15, P = 60
]


def t(a, dev):
    return torch.tensor(a, device=dev)


def run_forward(c, dev, **kw):
    views = R.ViewBatch.from_cameras([cam.to(dev) for cam in c.cams])
    return R.forward_views(views, t(c.means, dev), t(c.feat, dev), t(c.opac, dev), t(c.scales, dev), t(c.quats, dev),
                           None, want_aux=True, **kw)


@pytest.mark.parametrize("kw", CASES, ids=lambda k: f"seed{k['seed']}")
@pytest.mark.parametrize("binned", [False, True], ids=["small", "binned"])
def test_forward_bit_exact(device, kw, binned):
    c = util.make_case(**kw)
    color, inv, radii, st, final_T, n_contrib = run_forward(c, device, force_binned=binned)
    geo = R.decode_geom(st)
    for v in range(len(c.cams)):
        o = util.oracle_forward(c, v)
        vis = o["radii"] > 0
        assert np.array_equal(radii[v].cpu().numpy(), o["radii"])
        assert np.array_equal(geo["depths"][v].cpu().numpy()[vis], o["depths"][vis])
        assert np.array_equal(geo["xy"][v].cpu().numpy()[vis], o["xy"][vis])
        assert np.array_equal(geo["conic_opacity"][v].cpu().numpy()[vis], o["conic_opacity"][vis])
        assert np.array_equal(n_contrib[v].cpu().numpy().astype(np.uint32), o["n_contrib"])
        assert np.array_equal(final_T[v].cpu().numpy(), o["final_T"])
        assert np.array_equal(color[v].cpu().numpy(), o["color"])
        assert np.array_equal(inv[v].cpu().numpy(), o["invdepth"])
        if binned:
            pl, rg, nr = R.export_lists(st)
            Rn = int(nr[v].item())
            assert Rn == o["R"]
            assert np.array_equal(rg[v].cpu().numpy().astype(np.uint32), o["ranges"])
            assert np.array_equal(pl[v, :Rn].cpu().numpy().astype(np.uint32), o["point_list"])


@pytest.mark.parametrize("kw", CASES, ids=lambda k: f"seed{k['seed']}")
@pytest.mark.parametrize("binned", [False, True], ids=["small", "binned"])
@pytest.mark.parametrize("aa", [False, True], ids=["noaa", "aa"])
def test_backward_vs_oracle(device, kw, binned, aa):
    c = util.make_case(**kw)
    dev = device
    views = R.ViewBatch.from_cameras([cam.to(dev) for cam in c.cams])
    args = (t(c.means, dev), t(c.feat, dev), t(c.opac, dev), t(c.scales, dev), t(c.quats, dev), None)
    color, inv, radii, st = R.forward_views(views, *args, antialiasing=aa, force_binned=binned)
    bg = torch.tensor([0.3, 0.5, 0.2], device=dev)
    g = R.backward_views(st, *args, t(c.dL_color, dev), t(c.dL_inv, dev), bg=bg, want_dfeatures=True)
    for v in range(len(c.cams)):
        o = util.oracle_forward(c, v, antialiasing=aa)
        b = util.oracle_backward(c, v, o, antialiasing=aa, bg=[0.3, 0.5, 0.2])
        util.assert_close("dL_dmeans3D", g["means3D"][v].cpu(), b["dL_dmeans3D"])
        util.assert_close("dL_dmeans2D", g["means2D"][v].cpu(), b["dL_dmeans2D"])
        util.assert_close("dL_dopacity", g["opacities"][v].cpu(), b["dL_dopacity"])
        util.assert_close("dL_dscales", g["scales"][v].cpu(), b["dL_dscales"])
        util.assert_close("dL_drotations", g["rotations"][v].cpu(), b["dL_drotations"])
        util.assert_close("dL_dcov3D", g["cov3D"][v].cpu(), b["dL_dcov3D"])
        util.assert_close("dL_dfeatures", g["features"][v].cpu(), b["dL_dcolors"])
    # consume-and-clear: a second backward on the same scratch gives the same answer
    g2 = R.backward_views(st, *args, t(c.dL_color, dev), t(c.dL_inv, dev), bg=bg, want_dfeatures=False)
    util.assert_close("repeat", g2["means3D"].cpu(), g["means3D"].cpu(), rtol=1e-4)


def test_autograd_single_view_api(device):
    """The reference's call shape: GaussianRasterizer(settings)(means3D=..., shs=(P,1,C), ...) -> (color, radii, invdepth)."""
    import math
    from diff_gaussian_rasterization_h36m import GaussianRasterizationSettings, GaussianRasterizer
    c = util.make_case(seed=7, W=160, H=128, scale_log=4.0, onehot=True)
    dev = device
    cam = c.cams[0].to(dev)
    rs = GaussianRasterizationSettings(image_height=c.H, image_width=c.W, tanfovx=math.tan(cam.FoVx * 0.5),
                                       tanfovy=math.tan(cam.FoVy * 0.5), bg=torch.zeros(3, device=dev), scale_modifier=1.0,
                                       viewmatrix=cam.world_view_transform, projmatrix=cam.full_proj_transform, sh_degree=0,
                                       campos=cam.camera_center, prefiltered=False, debug=False, antialiasing=False)
    rast = GaussianRasterizer(rs)
    means = t(c.means, dev).requires_grad_(True)
    m2d = torch.zeros_like(means, requires_grad=True)
    sc = t(c.scales, dev).requires_grad_(True)
    q = t(c.quats, dev).requires_grad_(True)
    op = t(c.opac, dev).requires_grad_(True)
    shs = t(c.feat, dev)[:, None, :]
    color, radii, invd = rast(means3D=means, means2D=m2d, opacities=op, shs=shs, scales=sc, rotations=q)
    assert color.shape == (17, c.H, c.W) and radii.shape == (c.P,) and invd.shape == (1, c.H, c.W)
    assert radii.dtype == torch.int32
    o = util.oracle_forward(c, 0)
    assert np.array_equal(color.detach().cpu().numpy(), o["color"])
    (color * t(c.dL_color[0], dev)).sum().backward()
    b = util.oracle_backward(c, 0, o, with_inv=False)
    util.assert_close("means", means.grad.cpu(), b["dL_dmeans3D"])
    util.assert_close("m2d", m2d.grad.cpu(), b["dL_dmeans2D"])
    util.assert_close("scales", sc.grad.cpu(), b["dL_dscales"])
    util.assert_close("rot", q.grad.cpu(), b["dL_drotations"])
    util.assert_close("op", op.grad.cpu(), b["dL_dopacity"])
    with pytest.raises(Exception):
        rast(means3D=means, means2D=m2d, opacities=op, scales=sc, rotations=q)          # neither shs nor colors
    with pytest.raises(Exception):
        rast(means3D=means, means2D=m2d, opacities=op, shs=shs, scales=sc)               # rotations missing
    vis = rast.markVisible(means.detach())
    assert vis.dtype == torch.bool and vis.all()


def test_full_size_properties(device):
    """BASELINE config 2 shape (P=17, C=17, 1000x1000, V=4): size-independent properties instead of the slow oracle."""
    dev = device
    c = util.make_case(seed=11, W=1000, H=1000, n_views=4, scale_log=3.0, rand_rot=False, opac=1.0, ring=5000.0, onehot=True)
    views = R.ViewBatch.from_cameras([cam.to(dev) for cam in c.cams])
    args = (t(c.means, dev), t(c.feat, dev), t(c.opac, dev), t(c.scales, dev), t(c.quats, dev), None)
    col_s, inv_s, rad_s, st_s = R.forward_views(views, *args)
    col_b, inv_b, rad_b, st_b = R.forward_views(views, *args, force_binned=True)
    assert torch.equal(col_s, col_b) and torch.equal(inv_s, inv_b) and torch.equal(rad_s, rad_b)  # two code paths agree
    assert (rad_s > 0).all()
    assert float(col_s.min()) >= 0.0 and float(col_s.max()) <= 0.99 + 1e-6   # one-hot features, alpha <= 0.99
    # every joint paints only its own channel: off-channel mass is zero at the joint's pixel centre
    dL = torch.randn(4, 17, 1000, 1000, device=dev)
    g1 = R.backward_views(st_s, *args, dL)
    g2 = R.backward_views(st_s, *args, 2.0 * dL)
    util.assert_close("linearity", g2["means3D"].cpu(), 2.0 * g1["means3D"].cpu(), rtol=1e-4)
    gb = R.backward_views(st_b, *args, dL)
    util.assert_close("small-vs-binned", gb["means3D"].cpu(), g1["means3D"].cpu(), rtol=1e-4)
    # gradient lives only where the splats are: zeroing dL outside the touched tiles changes nothing
    mask = (col_s.sum(1, keepdim=True) > 0).float()
    mask = torch.nn.functional.max_pool2d(mask, 33, stride=1, padding=16)
    g3 = R.backward_views(st_s, *args, dL * mask)
    util.assert_close("support", g3["means3D"].cpu(), g1["means3D"].cpu(), rtol=1e-4)
