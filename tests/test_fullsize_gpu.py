"""Full-size oracle parity on every single-GPU BASELINE configuration, through the C ABI.

The sizes bench.py times are the sizes compared here with oracle/sks_oracle.c, view by view: BASELINE config 2 (H36M,
4 views @ 1000x1000, and the real sensor mix with 1002-wide cameras, dataset_readers.py:68-80), config 3 (Panoptic, all
31 views @ 1920x1080), config 5 (stress: P = 4 352, 8 views @ 2048^2, tile lists included) and the third shipped package
at its full shape (Occlusion-Person, 1280x720, C = 15, dataset_readers.py:350).  The C oracle needs 0.04 / 0.18 / 0.45 s
per view at these sizes; the size-dependent code paths (row-aligned vs linear fill, half-masked 16-byte stores at
W % 4 == 2, cover words beyond 8 at 1920 / 2048 wide, the composite-slot and backward workgroup splits that depend on
V x P) are therefore checked against the restatement itself, not through properties.

Bars: radii, colour planes, inverse depth, n_contrib, final_T, point_list, ranges: bit-exact
(np.array_equal).  The seven gradients: rtol 1e-3, atol 1e-5 x max|.| (the reference accumulates with fp32 atomics in
arbitrary order, backward.cu:593-635; the oracle sums in double).  Reference lines the oracle restates:
forward.cu:153-273, 278-401; backward.cu:147-449, 452-638; rasterizer_impl.cu:70-138.
"""
import math

import numpy as np
import pytest
import torch

from tests import util
from skelsplat_amd import rasterizer as R
from oracle import oracle as orc

pytestmark = pytest.mark.gpu

GRADS = (("means3D", "dL_dmeans3D"), ("means2D", "dL_dmeans2D"), ("opacities", "dL_dopacity"), ("scales", "dL_dscales"),
         ("rotations", "dL_drotations"), ("cov3D", "dL_dcov3D"), ("features", "dL_dcolors"))


def _ocam(cam):
    return orc.Cam(cam.image_width, cam.image_height, math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5),
                   cam.world_view_transform.cpu().numpy(), cam.full_proj_transform.cpu().numpy())


def _bench_scene(dataset, V, dev, W=None, H=None, seed=0):
    """The scene bench.py times (bench.make_scene): SyntheticScene + the GaussianModel's activated parameters
    (scaling 3 in log space, opacity exactly 1, identity rotations, one-hot features; gaussian_model.py:159-188)."""
    from skelsplat_amd.scene import SyntheticScene, GaussianModel
    sc = SyntheticScene(dataset, n_views=V, seed=seed, device=dev, W=W, H=H)
    gm = GaussianModel().create_from_points(sc.pose_3d_init, sc.spatial_lr_scale, sc.n_joints, scene_type=dataset, device=dev)
    P, C = sc.n_points, sc.n_joints
    with torch.no_grad():
        params = (gm.get_xyz.detach().clone(), gm.get_features.reshape(P, C).contiguous(), gm.get_opacity.detach().clone(),
                  gm.get_scaling.detach().clone(), gm.get_rotation.detach().clone())
    return sc.cameras, params, sc, gm


def _hard_case(dev, dataset, V, W, H, seed, ring, fx, scale_log=3.4, **kw):
    """Same cameras and skeleton recipe, but every parameter carries signal: anisotropic scales, random rotations,
    opacities in (0.3, 1), features one-hot + U(0, 0.1) on every channel (splats overlap and saturate)."""
    c = util.make_case(seed=seed, W=W, H=H, dataset=dataset, n_views=V, scale_log=scale_log, ring=ring,
                       fxmul=fx / (1145.0 * W / 1000.0), with_dL=False, **kw)
    t = lambda a: torch.tensor(a, device=dev)
    params = (t(c.means), t(c.feat), t(c.opac), t(c.scales), t(c.quats))
    return [cam.to(dev) for cam in c.cams], params


def _check_forward(cams, params, dev, binned_too=True, lists=False, bin_capacity=None, groups=None, aa=False):
    """All views, small and / or binned path, against the oracle.  Returns (oracle outputs per view, the forward state of
    the default path).  `groups`: lists of view indices that share an image size (the dense tensors cannot mix sizes)."""
    np_params = [p.detach().cpu().numpy() for p in params]
    P = params[0].shape[0]
    outs = [None] * len(cams)
    states = []
    for idx in (groups or [list(range(len(cams)))]):
        views = R.ViewBatch.from_cameras([cams[i] for i in idx])
        variants = [dict()] if P > 256 or not binned_too else [dict(), dict(force_binned=True)]
        for kw in variants:
            if P > 256 or kw:
                kw = dict(kw, bin_capacity=bin_capacity, check_capacity=True)
            color, inv, radii, st, final_T, n_contrib = R.forward_views(views, *params, None, want_aux=True, antialiasing=aa, **kw)
            if st.binning is not None and lists:
                pl, rg, nr = [x.cpu().numpy() for x in R.export_lists(st)]
            for k, i in enumerate(idx):
                o = outs[i] = outs[i] or orc.forward(*np_params, None, _ocam(cams[i]), antialiasing=aa)
                tag = f"view {i} {'binned' if st.binning is not None else 'small'}"
                assert np.array_equal(radii[k].cpu().numpy(), o["radii"]), tag
                assert np.array_equal(n_contrib[k].cpu().numpy().astype(np.uint32), o["n_contrib"]), tag
                assert np.array_equal(final_T[k].cpu().numpy(), o["final_T"]), tag
                assert np.array_equal(color[k].cpu().numpy(), o["color"]), tag
                assert np.array_equal(inv[k].cpu().numpy(), o["invdepth"]), tag
                if st.binning is not None and lists:
                    assert int(nr[k]) == o["R"], tag
                    assert np.array_equal(rg[k].astype(np.uint32), o["ranges"]), tag
                    assert np.array_equal(pl[k, :o["R"]].astype(np.uint32), o["point_list"]), tag
            if not kw or P > 256:
                states.append((idx, views, st))
            del color, inv, final_T, n_contrib
    return outs, states


def _check_backward(cams, params, dev, outs, states, seed=0, rtol=1e-3, bg=None, binned_too=True, aa=False):
    """Dense random dL/d(colour) and dL/d(inverse depth) on the GPU, every view's seven gradients against the oracle."""
    np_params = [p.detach().cpu().numpy() for p in params]
    P, C = params[1].shape
    gen = torch.Generator(device=dev).manual_seed(seed)
    for idx, views, st in states:
        H, W = views.H, views.W
        dLc = torch.randn((len(idx), C, H, W), device=dev, generator=gen)
        dLi = torch.randn((len(idx), 1, H, W), device=dev, generator=gen)
        bgt = None if bg is None else torch.tensor(bg, device=dev)
        todo = [("default", st)]
        if binned_too and st.binning is None:
            stb = R.forward_views(views, *params, None, force_binned=True, check_capacity=True, antialiasing=aa)[3]
            todo.append(("binned", stb))
        for name, s in todo:
            g = R.backward_views(s, *params, None, dLc, dLi, bg=bgt, want_dfeatures=True)
            g = {k: v.cpu().numpy() for k, v in g.items() if v is not None}
            for k, i in enumerate(idx):
                if name == "default":
                    outs[i]["_bwd"] = orc.backward(outs[i], *np_params, None, _ocam(cams[i]), dLc[k].cpu().numpy(),
                                                   dLi[k].cpu().numpy(), bg=bg, antialiasing=aa)
                b = outs[i]["_bwd"]
                for ours, theirs in GRADS:
                    util.assert_close(f"view {i} {name} {theirs}", g[ours][k], b[theirs].reshape(g[ours][k].shape), rtol=rtol)
        del dLc, dLi
    for o in outs:
        o.pop("_bwd", None)


def _check_fused_loss(cams, params, dev, outs, hm_planes):
    """The sparse fused training step (sks_geometry + sks_backward_fused_loss: render, clamp(0, 1), masked L2 and its
    gradient on the covered tiles only; train.py:140-161, gaussian_renderer/__init__.py:129, loss_utils.py:86-100) against
    oracle render -> clamp -> masked L2 -> oracle backward on dense images.  hm_planes[v]: (C,H_v,W_v) device tensor."""
    np_params = [p.detach().cpu().numpy() for p in params]
    P, C = params[1].shape
    views = R.ViewBatch.from_cameras(cams, allow_mixed=True)
    if views.mixed:      # views of different sizes: one flat buffer + per-view offsets
        hs = R.HeatmapSet(views.sizes, C, dev)
        for v, pl in enumerate(hm_planes):
            hs.planes[v].copy_(pl)
        stats = R.GtStats()
        stats.gt, stats.offsets, stats.tile_S, stats.tile_N = hs.flat, hs.offsets, None, None
        stats.totals = torch.empty((views.V, 2), dtype=torch.float64, device=dev)
        for (w, h), vs in hs.groups.items():
            stats.totals[vs] = R.gt_tile_stats(hs.group((w, h))).totals
    else:
        stats = R.gt_tile_stats(torch.stack(list(hm_planes)))
    st = R.geometry_views(views, params[0], C, params[2], params[3], params[4], None)
    g, sums = R.backward_fused_loss(st, stats, *params, None)
    g = {k: v.cpu().numpy() for k, v in g.items() if v is not None}
    sums = sums.cpu().numpy()
    for v, cam in enumerate(cams):
        o = outs[v]
        gt = hm_planes[v].cpu().numpy()
        render = np.clip(o["color"], 0.0, 1.0)
        mask = (gt > 0) | (render > 0)
        diff = (render - gt).astype(np.float32)
        S = float((diff.astype(np.float64) ** 2)[mask].sum())
        N = int(mask.sum())
        assert int(sums[v, 1]) == N, (v, sums[v, 1], N)                     # mask counts are integers: exact
        assert abs(sums[v, 0] - S) <= 1e-5 * S, (v, sums[v, 0], S)
        dL = (2.0 * diff * mask * ((o["color"] >= 0) & (o["color"] <= 1))).astype(np.float32)   # clamp's pass-through
        b = orc.backward(o, *np_params, None, _ocam(cam), dL, None)
        for ours, theirs in GRADS[:5]:
            util.assert_close(f"fused view {v} {theirs}", g[ours][v], b[theirs].reshape(g[ours][v].shape), rtol=1e-3)
        assert np.abs(b["dL_dmeans3D"]).max() > 0


def _heatmaps(sc, gm, dev, cams=None, p2d=None):
    from skelsplat_amd.heatmaps import generate_heatmaps
    return generate_heatmaps(gm._xyz.detach(), gm.get_scaling.detach(), gm._rotation.detach(),
                             torch.tensor(sc.poses_2d if p2d is None else p2d, device=dev), cams or sc.cameras)


# ------------------------------------------------------------------------------------------------------------
# BASELINE config 2: H36M, P = C = 17, 4 views @ 1000x1000
# ------------------------------------------------------------------------------------------------------------
def test_config2_h36m_bench_scene(device):
    """Exactly the scene bench.py's headline times (seed 0), all four views: forward bit-exact (small and binned path,
    lists included), backward on both paths, the sparse fused-loss step."""
    cams, params, sc, gm = _bench_scene("h36m", 4, device)
    outs, states = _check_forward(cams, params, device, lists=True)
    assert all((o["n_contrib"] > 0).sum() > 3000 for o in outs)
    _check_backward(cams, params, device, outs, states)
    hm = _heatmaps(sc, gm, device)
    _check_fused_loss(cams, params, device, outs, [hm[v] for v in range(4)])


def test_config2_h36m_hard_parameters(device):
    """Config 2's shape with parameters that exercise every term: anisotropic rotated covariances, finite opacities,
    non-one-hot features, antialiasing off; a non-zero background in the backward (Q2)."""
    cams, params = _hard_case(device, "h36m", 4, 1000, 1000, seed=21, ring=5000.0, fx=1145.0)
    outs, states = _check_forward(cams, params, device, lists=True)
    assert all((o["n_contrib"] > 1).sum() > 2000 for o in outs)           # overlapping splats
    _check_backward(cams, params, device, outs, states, bg=[0.3, 0.5, 0.2])
    # pipe.antialiasing = True (forward.cu:226-227, backward.cu:210-246): the opacity scaling and its gradient
    outs, states = _check_forward(cams[:2], params, device, aa=True)
    _check_backward(cams[:2], params, device, outs, states, aa=True)


def test_config2_h36m_sensor_mix_1002(device):
    """H36M's real sensor mix (quirk Q11): 1002x1000 and 1000x1000 cameras in one accumulation group -- the dense API
    takes one size per call, the sparse fused step takes the mixed group in one launch sequence."""
    from skelsplat_amd.scene import SyntheticScene, GaussianModel, Camera
    sc = SyntheticScene("h36m", n_views=4, seed=3, device=device)
    cams = []
    for v, cam in enumerate(sc.cameras):
        W = 1002 if v in (0, 3) else 1000
        K = cam.K.copy()
        K[0, 2] += (W - 1000) / 2
        cams.append(Camera(cam.uid, cam.R, cam.T, K, W, 1000, device=device))
    gm = GaussianModel().create_from_points(sc.pose_3d_init, sc.spatial_lr_scale, 17, device=device)
    with torch.no_grad():
        params = (gm.get_xyz.detach().clone(), gm.get_features.reshape(17, 17).contiguous(), gm.get_opacity.detach().clone(),
                  gm.get_scaling.detach().clone(), gm.get_rotation.detach().clone())
    groups = [[0, 3], [1, 2]]
    outs, states = _check_forward(cams, params, device, groups=groups)
    _check_backward(cams, params, device, outs, states)
    hm = [_heatmaps(sc, gm, device, [cams[v]], sc.poses_2d[v:v + 1])[0] for v in range(4)]
    assert hm[0].shape == (17, 1000, 1002) and hm[1].shape == (17, 1000, 1000)
    _check_fused_loss(cams, params, device, outs, hm)
    # and the 1002-wide sensor with parameters that overlap and saturate
    cams2, params2 = _hard_case(device, "h36m", 2, 1002, 1000, seed=23, ring=5000.0, fx=1145.0)
    outs2, states2 = _check_forward(cams2, params2, device)
    _check_backward(cams2, params2, device, outs2, states2)


# ------------------------------------------------------------------------------------------------------------
# BASELINE config 3: Panoptic, P = C = 19, all 31 views @ 1920x1080 in one launch
# ------------------------------------------------------------------------------------------------------------
def test_config3_panoptic_all_31_views(device):
    cams, params, sc, gm = _bench_scene("panoptic", 31, device)
    outs, states = _check_forward(cams, params, device)
    assert all((o["n_contrib"] > 0).sum() > 10000 for o in outs)
    _check_backward(cams, params, device, outs, states)
    torch.cuda.empty_cache()
    hm = _heatmaps(sc, gm, device)
    _check_fused_loss(cams, params, device, outs, [hm[v] for v in range(31)])


def test_config3_panoptic_hard_parameters(device):
    cams, params = _hard_case(device, "panoptic", 6, 1920, 1080, seed=25, ring=3000.0, fx=1400.0)
    outs, states = _check_forward(cams, params, device, lists=True)
    assert all((o["n_contrib"] > 1).sum() > 5000 for o in outs)
    _check_backward(cams, params, device, outs, states, bg=[0.1, 0.0, 0.7])


# ------------------------------------------------------------------------------------------------------------
# BASELINE config 5: 256 skeletons (P = 4 352, C = 17), 8 views @ 2048x2048, binned path
# ------------------------------------------------------------------------------------------------------------
def _stress(dev, V, **kw):
    return _hard_case(dev, "h36m", V, 2048, 2048, seed=42, ring=20000.0, fx=2300.0, scale_log=3.0, n_skeletons=256,
                      pitch=1500.0, **kw)


def test_config5_stress_bench_scene(device):
    """The scene tools/bench_stress.py and bench.py's `stress` extra time (one-hot features, opacity 1): 8 views,
    forward + tile lists bit-exact, all gradients."""
    cams, params = _stress(device, 8, onehot=True, opac=1.0)
    assert params[0].shape[0] == 4352
    outs, states = _check_forward(cams, params, device, lists=True, bin_capacity=400000)
    assert all(o["R"] > 4000 for o in outs)
    _check_backward(cams, params, device, outs, states)


def test_config5_stress_hard_parameters(device):
    cams, params = _stress(device, 3)
    outs, states = _check_forward(cams, params, device, lists=True)
    _check_backward(cams, params, device, outs, states, bg=[0.3, 0.5, 0.2])


# ------------------------------------------------------------------------------------------------------------
# Occlusion-Person: C = 15 @ 1280x720 (the third shipped package at its full shape)
# ------------------------------------------------------------------------------------------------------------
def test_occlusion_person_full_size(device):
    cams, params, sc, gm = _bench_scene("occlusion-person", 8, device)
    assert params[1].shape == (15, 15) and (cams[0].image_width, cams[0].image_height) == (1280, 720)
    outs, states = _check_forward(cams, params, device, lists=True)
    _check_backward(cams, params, device, outs, states)
    hm = _heatmaps(sc, gm, device)
    _check_fused_loss(cams, params, device, outs, [hm[v] for v in range(8)])
    cams2, params2 = _hard_case(device, "occlusion-person", 3, 1280, 720, seed=27, ring=5000.0, fx=1145.0)
    outs2, states2 = _check_forward(cams2, params2, device)
    _check_backward(cams2, params2, device, outs2, states2, bg=[0.3, 0.5, 0.2])
