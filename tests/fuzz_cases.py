"""One randomised case of the binned path against the CPU oracle (shared by tests/test_fuzz_gpu.py and tools/fuzz_binned.py): image
sizes that are not multiples of the tile, 1-3 views, 1-40 skeletons at random pitch and scale (tile lists from one entry to several
hundred), one-hot or dense features, opacities below 1, clamp, antialiasing, a background, with and without the inverse-depth and the
feature gradient.  Forward bit for bit (colour, inverse depth, contributor counts, final T, tile lists), gradients at the tests'
tolerance."""
import numpy as np
import torch

from tests import util
from oracle import oracle as orc_mod
from skelsplat_amd import rasterizer as R

GR = (("means3D", "dL_dmeans3D"), ("means2D", "dL_dmeans2D"), ("opacities", "dL_dopacity"), ("scales", "dL_dscales"),
      ("rotations", "dL_drotations"), ("cov3D", "dL_dcov3D"), ("features", "dL_dcolors"))


def run_case(seed, dev, small_path_too=False, stats=None):
    """Raises AssertionError (its text names the case) when anything differs.  (check_capacity=True: the raw entry point's default
    sizes the binning arena once per shape and checks later calls lazily -- two random scenes of one shape would trip it.)
    `stats`: a dict that receives, per gradient, the largest excess over rtol in units of 2^-24 x sum|terms| (util.bound_excess:
    what tools/fuzz_bound_calib.py calibrates util.BOUND_KAPPA with)."""
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev) if a is not None else None
    rng = np.random.default_rng(seed)
    W, H = int(rng.integers(40, 260)), int(rng.integers(40, 200))
    if seed % 8 == 0:    # wide and flat: the row-aligned fill mode (W % 32 == 0, >= 90 % of a 1024-pixel chunk) and its neighbours
        W, H = int(rng.choice([928, 960, 1000, 1002, 1024, 1056, 1990, 2048])), int(rng.integers(20, 70))
    nv = int(rng.integers(1, 4))
    nsk = int(rng.choice([1, 2, 6, 18, 40]))
    kw = dict(seed=seed, W=W, H=H, n_views=nv, n_skeletons=nsk, scale_log=float(rng.uniform(2.5, 4.6)),
              pitch=float(rng.uniform(20.0, 700.0)), onehot=bool(rng.integers(0, 2)),
              opac=None if rng.integers(0, 2) else float(rng.choice([0.05, 0.3, 0.6, 1.0])), fxmul=float(rng.uniform(0.6, 1.6)))
    aa, clamp = bool(rng.integers(0, 2)), bool(rng.integers(0, 3) == 0)
    use_bg, use_inv, use_feat = bool(rng.integers(0, 2)), bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
    smod = float(rng.choice([1.0, 1.0, 1.25, 0.7]))             # raster_settings.scale_modifier
    precomp = bool(rng.integers(0, 4) == 0)                      # pipe.compute_cov3D_python: six numbers per Gaussian, no scales / rotations
    c = util.make_case(**kw)
    extreme = seed % 5 == 0
    if extreme:    # Gaussians behind a camera / far off screen, pin-point and image-sized, transparent and opaque, flat as a sheet
        P = c.P
        pick = lambda frac: rng.random(P) < frac
        c.means = c.means.copy(); c.scales = c.scales.copy(); c.opac = c.opac.copy()
        m = pick(0.08); c.means[m] += rng.normal(0.0, 4000.0, (int(m.sum()), 3)).astype(np.float32)      # anywhere, also behind the cameras
        m = pick(0.06); c.scales[m] *= 1e-3                                                                 # sub-pixel
        m = pick(0.04); c.scales[m] *= 30.0                                                                 # covers the image
        m = pick(0.05); c.scales[m, rng.integers(0, 3, int(m.sum()))] *= 1e-4                               # flat
        m = pick(0.05); c.opac[m] = 0.0
        m = pick(0.05); c.opac[m] = 1.0
        m = pick(0.03); c.opac[m] = 1.0 / 255.0
        c.feat = c.feat.copy()
        m = pick(0.06); c.feat[m] = 0.0                                                                      # no colour at all: only inverse depth
    views = R.ViewBatch.from_cameras([cam.to(dev) for cam in c.cams])
    cov = None
    if precomp:
        cov = orc_mod.forward(c.means, c.feat, c.opac, c.scales, c.quats, None, c.ocams[0], scale_modifier=smod)["cov3D"].astype(np.float32)
    o_args = (c.means, c.feat, c.opac, None, None, cov) if precomp else (c.means, c.feat, c.opac, c.scales, c.quats, None)
    args = tuple(t(x) for x in o_args)
    tag = (f"seed {seed}: {W}x{H} V={nv} P={c.P} aa={aa} clamp={clamp} bg={use_bg} inv={use_inv} feat={use_feat} "
           f"scale_modifier={smod} cov3D_precomp={precomp} extreme={extreme} {kw}")
    try:
        col, inv, radii, st, fT, nC = R.forward_views(views, *args, scale_modifier=smod, antialiasing=aa, want_aux=True, force_binned=True, check_capacity=True)
        pl, rg, nr = R.export_lists(st)
        outs = []
        for v in range(nv):
            o = orc_mod.forward(*o_args, c.ocams[v], scale_modifier=smod, antialiasing=aa)
            outs.append(o)
            assert np.array_equal(radii[v].cpu().numpy(), o["radii"]), "radii"
            assert np.array_equal(col[v].cpu().numpy(), o["color"]), "color"
            assert np.array_equal(inv[v, 0].cpu().numpy(), o["invdepth"].reshape(H, W)), "invdepth"
            assert np.array_equal(nC[v].cpu().numpy().astype(np.uint32), o["n_contrib"]), "n_contrib"
            assert np.array_equal(fT[v].cpu().numpy(), o["final_T"]), "final_T"
            assert int(nr[v]) == o["R"] and np.array_equal(rg[v].cpu().numpy(), o["ranges"]), "ranges"
            assert np.array_equal(pl[v, :o["R"]].cpu().numpy(), o["point_list"]), "point_list"
        if clamp:
            col2, _, _, st = R.forward_views(views, *args, scale_modifier=smod, antialiasing=aa, clamp01=True, force_binned=True, check_capacity=True)
            for v in range(nv):
                assert np.array_equal(col2[v].cpu().numpy(), np.clip(outs[v]["color"], 0.0, 1.0)), "clamped color"
        bg = [0.3, 0.1, 0.7] + [0.0] * (c.C - 3) if use_bg else None
        bgt = None if bg is None else torch.tensor(bg, device=dev)
        g = R.backward_views(st, *args, t(c.dL_color), t(c.dL_inv) if use_inv else None, bg=bgt, want_dfeatures=use_feat)
        obw = []
        if not clamp:   # (the oracle's backward has no clamp; the clamped path is held to torch autograd in tests/)
            for v in range(nv):
                b = orc_mod.backward(outs[v], *o_args, c.ocams[v], c.dL_color[v], c.dL_inv[v] if use_inv else None, bg=bg,
                                     scale_modifier=smod, antialiasing=aa, bounds=extreme)
                obw.append(b)
                for ours, theirs in GR:
                    if g.get(ours) is None or (precomp and ours in ("scales", "rotations")):
                        continue
                    got, want = g[ours][v].cpu().numpy(), b[theirs].reshape(g[ours][v].shape)
                    if extreme:
                        # an image-sized needle's gradients are sums of thousands of pixel terms that cancel to 1e-5 of their
                        # size, a sheet's quaternion gradient a difference of terms 1e8 apart: no fraction of the tensor's
                        # largest entry is the right allowance there.  The oracle computes one: beside every sum the sum of
                        # |terms|, through the geometry backward on absolute values (oracle.backward(bounds=True)); two fp32
                        # evaluations may differ by util.BOUND_KAPPA x 2^-24 of that, however small the gradient comes out --
                        # a missing tile is four orders of magnitude above it
                        util.assert_close_bound(f"{theirs} view {v}", got, want, b["bound"][theirs].reshape(got.shape), rtol=1e-3)
                        if stats is not None:
                            stats[theirs] = max(stats.get(theirs, 0.0), util.bound_excess(got, want, b["bound"][theirs].reshape(got.shape)))
                    else:
                        util.assert_close(f"{theirs} view {v}", got, want, rtol=1e-3, atol_scale=1e-5)
        if c.P <= 256:   # the small path (fill + sparse composite, wave-resident / gather backward) on the same case
            col, inv, radii, st, fT, nC = R.forward_views(views, *args, scale_modifier=smod, antialiasing=aa, want_aux=True, clamp01=clamp)
            for v in range(nv):
                o = outs[v]
                assert np.array_equal(col[v].cpu().numpy(), np.clip(o["color"], 0.0, 1.0) if clamp else o["color"]), "small: color"
                assert np.array_equal(inv[v, 0].cpu().numpy(), o["invdepth"].reshape(H, W)), "small: invdepth"
                assert np.array_equal(nC[v].cpu().numpy().astype(np.uint32), o["n_contrib"]), "small: n_contrib"
                assert np.array_equal(fT[v].cpu().numpy(), o["final_T"]), "small: final_T"
            g2 = R.backward_views(st, *args, t(c.dL_color), t(c.dL_inv) if use_inv else None, bg=bgt, want_dfeatures=use_feat)
            for ours, theirs in GR:   # the two paths against each other (clamped or not)
                if g.get(ours) is None or g2.get(ours) is None:
                    continue
                if extreme and obw:       # (both are fp32 evaluations of the oracle's sums: twice its rounding allowance)
                    for v in range(nv):
                        util.assert_close_bound(f"small vs binned {theirs} view {v}", g2[ours][v].cpu().numpy(), g[ours][v].cpu().numpy(),
                                                obw[v]["bound"][theirs].reshape(g[ours][v].shape), rtol=1e-3, kappa=2 * util.BOUND_KAPPA)
                elif not extreme:
                    util.assert_close(f"small vs binned {theirs}", g2[ours].cpu().numpy(), g[ours].cpu().numpy(), rtol=1e-3, atol_scale=1e-5)
                # (extreme AND clamped: the oracle has no clamped backward to take a bound from; the un-clamped extreme cases and
                # the clamped ordinary ones cover the two switches)
    except AssertionError as e:
        raise AssertionError(f"{tag} -> {str(e)[:300]}") from None


def run_one_call_case(seed, dev):
    """sks_forward_backward (the backward on a second stream beside the forward, joined or not) against sks_forward + sks_backward on a
    random small-path scene: image, inverse depth, radii and every gradient BIT FOR BIT, over several calls on one Workspace with
    the parameters changing in place in between, any switch combination, 1-6 views, odd image sizes; every third seed with more
    than SKS_MAX_CHANNELS channels (the generic path: two calls under the hood, same bits)."""
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev) if a is not None else None
    rng = np.random.default_rng(seed)
    W, H = int(rng.integers(40, 300)), int(rng.integers(40, 220))
    if seed % 6 == 0:
        W, H = int(rng.choice([1000, 1002, 1024, 1920])), int(rng.integers(20, 60))
    nv = int(rng.integers(1, 7))
    c = util.make_case(seed=seed, W=W, H=H, n_views=nv, n_skeletons=int(rng.choice([1, 1, 2, 6])), scale_log=float(rng.uniform(2.8, 4.4)),
                       pitch=float(rng.uniform(60.0, 600.0)), onehot=bool(rng.integers(0, 2)), fxmul=float(rng.uniform(0.7, 1.5)))
    aa, clamp = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
    use_bg, use_inv, use_feat, want_mean = (bool(rng.integers(0, 2)) for _ in range(4))
    feat = c.feat
    if seed % 3 == 0:
        Cw = int(rng.integers(33, 72))
        feat = (rng.random((c.P, Cw)) * (rng.random((c.P, Cw)) < 0.4)).astype(np.float32)
        want_mean = False
    C = feat.shape[1]
    tag = f"one-call seed {seed}: {W}x{H} V={nv} P={c.P} C={C} aa={aa} clamp={clamp} bg={use_bg} inv={use_inv} feat={use_feat} mean={want_mean}"
    try:
        views = R.ViewBatch.from_cameras([cam.to(dev) for cam in c.cams])
        means, fe, opac, scales, quats = (t(a) for a in (c.means, feat, c.opac, c.scales, c.quats))
        dL = torch.randn((nv, C, H, W), device=dev, generator=torch.Generator(device=dev).manual_seed(seed))
        dLi = torch.randn((nv, 1, H, W), device=dev, generator=torch.Generator(device=dev).manual_seed(seed + 1)) if use_inv else None
        bg = torch.rand(C, device=dev) if use_bg else None
        kw = dict(antialiasing=aa, clamp01=clamp, bg=bg, want_dfeatures=use_feat, want_mean=want_mean)
        ws = R.Workspace()
        for rep in range(4):
            if rep == 2:
                with torch.no_grad():
                    means.add_(float(rng.uniform(-4.0, 4.0)))
                    scales.mul_(float(rng.uniform(0.9, 1.1)))
                    dL.mul_(-0.5)
            col, inv, rad, st = R.forward_views(views, means, fe, opac, scales, quats, None, antialiasing=aa, clamp01=clamp)
            g = R.backward_views(st, means, fe, opac, scales, quats, None, dL, dLi, bg=bg, want_dfeatures=use_feat, want_mean=want_mean)
            join = bool((seed + rep) % 2)
            out = R.forward_backward_views(views, means, fe, opac, scales, quats, None, dL, dLi, workspace=ws, join=join, **kw)
            if not join:
                ws.join(dev.index)
            assert torch.equal(col, out[0]) and torch.equal(inv, out[1]) and torch.equal(rad, out[2]), ("forward", rep)
            for k, v in g.items():
                assert (v is None and out[4][k] is None) or torch.equal(v, out[4][k]), (k, rep)
        return dict(visible=float((rad > 0).sum()), grad=float(g["means3D"].abs().max()))
    except AssertionError as e:
        raise AssertionError(f"{tag} -> {str(e)[:300]}") from None


def run_fused_loss_case(seed, dev):
    """The production loop's step -- sks_geometry + sks_backward_fused_loss: no image, no dense gradient, the pseudo-GT as planes
    or as separable factors -- against the dense device path sks_forward(clamp) -> sks_masked_l2 -> sks_backward on a random
    scene: odd image sizes, 1-5 views, the skeleton displaced from its pseudo-GT, random Gaussian and heat-map scales."""
    from skelsplat_amd.ops import masked_l2
    from skelsplat_amd.scene import SyntheticScene, GaussianModel
    from skelsplat_amd.heatmaps import generate_heatmaps, heatmap_factors
    rng = np.random.default_rng(seed)
    dataset = str(rng.choice(["h36m", "panoptic", "occlusion-person"]))
    W, H = int(rng.integers(48, 300)), int(rng.integers(48, 220))
    nv = int(rng.integers(1, 6))
    scaling, hscale = float(rng.uniform(2.6, 4.6)), float(rng.uniform(0.7, 1.6))
    fxm, shift = float(rng.uniform(0.7, 1.8)), float(rng.uniform(0.0, 60.0))
    tag = f"fused-loss seed {seed}: {dataset} {W}x{H} V={nv} scaling={scaling:.2f} heat x{hscale:.2f} fx x{fxm:.2f} shift {shift:.0f} mm"
    try:
        sc = SyntheticScene(dataset, n_views=nv, seed=seed, W=W, H=H, ring=2500.0, fx=1145.0 * (W / 1000) * fxm, device=dev)
        init = sc.pose_3d_init + rng.normal(0.0, shift, sc.pose_3d_init.shape)
        gm = GaussianModel().create_from_points(init, sc.spatial_lr_scale, sc.n_joints, scaling=scaling, scene_type=dataset, device=dev)
        gt3d = torch.tensor(sc.pose_3d_gt, device=dev).float()
        p2d = torch.tensor(sc.poses_2d, device=dev)
        hm = generate_heatmaps(gt3d, gm.get_scaling.detach() * hscale, gm._rotation.detach(), p2d, sc.cameras)
        P, C = sc.n_points, sc.n_joints
        with torch.no_grad():
            args = (gm._xyz.detach(), gm.get_features.reshape(P, C), gm.get_opacity.detach(), gm.get_scaling.detach(),
                    gm.get_rotation.detach(), None)
        views = R.ViewBatch.from_cameras(sc.cameras)
        color, inv, radii, st = R.forward_views(views, *args, clamp01=True)
        dL, S, N = masked_l2(color, hm)
        gd = R.backward_views(st, *args, dL)
        stats = R.gt_tile_stats(hm)
        st2 = R.geometry_views(views, args[0], C, args[2], args[3], args[4], None)
        gs, sums = R.backward_fused_loss(st2, stats, *args)
        assert torch.equal(st2.radii, radii), "radii"
        assert torch.equal(sums[:, 1], N), ("mask counts", sums[:, 1], N)
        assert ((sums[:, 0] - S).abs() <= 1e-5 * S.abs() + 1e-12).all(), ("loss sums", sums[:, 0], S)
        for k in ("means3D", "means2D", "opacities", "scales", "rotations"):
            util.assert_close(k, gs[k].cpu(), gd[k].cpu(), rtol=1e-4, atol_scale=1e-5)
        fac = R.HeatmapFactors(nv, C, W, H, dev)
        heatmap_factors(gt3d, gm.get_scaling.detach() * hscale, gm._rotation.detach(), p2d, sc.cameras, views=views, out=fac)
        for v in range(nv):
            assert torch.equal(fac.planes(v), hm[v]), "factor planes"
        fst = R.GtStats()
        fst.gt, fst.tile_S, fst.tile_N, fst.factors = None, None, None, fac
        fst.totals = fac.totals(views, torch.empty((nv, 2), dtype=torch.float64, device=dev))
        assert torch.equal(fst.totals[:, 1], stats.totals[:, 1]), "factor totals N"
        gf, sums_f = R.backward_fused_loss(st2, fst, *args)
        assert torch.equal(sums_f[:, 1], sums[:, 1]), "factor mask counts"
        for k in ("means3D", "means2D", "opacities", "scales", "rotations"):
            assert torch.equal(gf[k], gs[k]), f"factors vs planes {k}"
        return dict(mask_pixels=float(N.min()), grad=float(gd["means3D"].abs().max()))   # (how much the case exercised)
    except AssertionError as e:
        raise AssertionError(f"{tag} -> {str(e)[:300]}") from None


def run_loop_case(seed, dev):
    """MultiViewLoop's production path (sparse fused step, device-side Adam tail, hipGraph) against its dense per-launch path (full
    images, sks_masked_l2, dense backward, the same optimiser) on a random scene: 1-6 views under an accumulation_steps of 1-5 (quirk
    Q8: V != accumulation_steps), odd image sizes, the three datasets, antialiasing, opacity on or off, 8-40 iterations."""
    from skelsplat_amd.loop import MultiViewLoop
    from skelsplat_amd.scene import SyntheticScene, GaussianModel
    from skelsplat_amd.heatmaps import generate_heatmaps
    rng = np.random.default_rng(seed)
    dataset = str(rng.choice(["h36m", "panoptic", "occlusion-person"]))
    W, H = int(rng.integers(64, 240)), int(rng.integers(64, 200))
    nv, acc = int(rng.integers(1, 7)), int(rng.integers(1, 6))
    iters = int(rng.integers(2, 9)) * acc + int(rng.integers(0, acc))
    scaling, fxm = float(rng.uniform(3.0, 4.4)), float(rng.uniform(0.9, 1.7))
    aa, op_on = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
    lam = float(rng.choice([0.0, 1e-5, 1e-2]))
    tag = f"loop seed {seed}: {dataset} {W}x{H} V={nv} acc={acc} iterations={iters} scaling={scaling:.2f} aa={aa} opacity={op_on} lambda={lam}"
    try:
        sc = SyntheticScene(dataset, n_views=nv, seed=seed, W=W, H=H, ring=2500.0, fx=1145.0 * (W / 1000) * fxm, device=dev)
        cams = sc.cameras
        mixed = nv > 1 and seed % 3 == 0     # two sensor widths in one accumulation group (H36M's 1000 / 1002, quirk Q11)
        if mixed:
            from skelsplat_amd.scene import Camera
            cams = []
            for v, cam in enumerate(sc.cameras):
                Wv = W + (2 if v % 2 else 0)
                K = cam.K.copy()
                K[0, 2] += (Wv - W) / 2
                cams.append(Camera(cam.uid, cam.R, cam.T, K, Wv, H, device=dev))
            tag += " mixed widths"
        outs = []
        for sparse in (True, False):
            gm = GaussianModel().create_from_points(sc.pose_3d_init, sc.spatial_lr_scale, sc.n_joints, scaling=scaling,
                                                    opacity_on=op_on, scene_type=dataset, device=dev)
            gm.training_setup()
            p2d = torch.tensor(sc.poses_2d, device=dev)
            if mixed:
                hm = [generate_heatmaps(gm._xyz.detach(), gm.get_scaling.detach(), gm._rotation.detach(), p2d[v:v + 1], [cams[v]])[0]
                      for v in range(nv)]
            else:
                hm = generate_heatmaps(gm._xyz.detach(), gm.get_scaling.detach(), gm._rotation.detach(), p2d, cams)
            loop = MultiViewLoop(gm, cams, hm, dataset=dataset, accumulation_steps=acc, lambda_consistency=lam,
                                 antialiasing=aa, sparse=sparse, use_graph=sparse)
            assert loop.sparse == sparse
            loop.run(iters)
            outs.append((gm._xyz.detach().cpu().clone(), gm._scaling.detach().cpu().clone(), gm._opacity.detach().cpu().clone()))
        init = torch.tensor(sc.pose_3d_init).float()
        moved = (outs[1][0] - init).norm(dim=1).mean().item()
        diff = (outs[0][0] - outs[1][0]).norm(dim=1).max().item()
        assert diff <= 2e-3 * max(moved, 0.05) + 1e-4, ("joints", diff, moved)
        util.assert_close("scaling", outs[0][1], outs[1][1], rtol=1e-4, atol_scale=1e-4)
        fin = torch.isfinite(outs[1][2])
        assert torch.equal(fin, torch.isfinite(outs[0][2])), "opacity finiteness"
        if fin.any():
            util.assert_close("opacity", outs[0][2][fin], outs[1][2][fin], rtol=1e-4, atol_scale=1e-4)
        return dict(moved_mm=moved, steps=iters // acc)
    except AssertionError as e:
        raise AssertionError(f"{tag} -> {str(e)[:300]}") from None


def run_frames_case(seed, dev):
    """FrameBatchLoop (F frames per launch, planes or factors, hipGraph or not) against MultiViewLoops running the frames one by
    one: every frame's parameters, Adam moments, V-slot buffers and counters BIT FOR BIT, on a random scene (1-8 frames, 1-9 views,
    odd image sizes, the three datasets, dropped heat-map planes)."""
    from skelsplat_amd.loop import MultiViewLoop, FrameBatchLoop
    from skelsplat_amd.scene import SyntheticScene, GaussianModel
    from skelsplat_amd.heatmaps import generate_heatmaps
    rng = np.random.default_rng(seed)
    dataset = str(rng.choice(["h36m", "panoptic", "occlusion-person"]))
    W, H = int(rng.integers(64, 220)), int(rng.integers(64, 180))
    F, V = int(rng.integers(1, 9)), int(rng.integers(1, 10))
    F = min(F, 64 // V)          # (SKS_MAX_VIEWS views per launch)
    iters = 4 * int(rng.integers(1, 8)) + int(rng.integers(0, 4))
    use_graph, factored = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
    gpg = int(rng.integers(1, 5))
    tag = f"frames seed {seed}: {dataset} {W}x{H} F={F} V={V} iterations={iters} graph={use_graph}/{gpg} factored={factored}"
    try:
        sc = SyntheticScene(dataset, n_views=V, seed=seed, W=W, H=H, ring=2500.0, fx=1145.0 * (W / 1000) * 1.5, device=dev)

        def model():
            gm = GaussianModel().create_from_points(sc.pose_3d_init, sc.spatial_lr_scale, sc.n_joints, scaling=3.9,
                                                    scene_type=dataset, device=dev)
            gm.training_setup()
            return gm
        cams, J = sc.cameras, sc.n_joints
        base3, base2 = np.asarray(sc.pose_3d_init, np.float32), np.asarray(sc.poses_2d, np.float32)
        pts = np.stack([base3 + rng.normal(0, 25.0 * (f + 1), base3.shape) for f in range(F)]).astype(np.float32)
        p2d = np.stack([base2 + rng.normal(0, 2.0 * (f + 1), base2.shape) for f in range(F)]).astype(np.float32)
        drop = torch.tensor(rng.random((F, V, J)) < 0.04)
        fb = FrameBatchLoop(model(), cams, F, dataset=dataset, use_graph=use_graph, factored=factored)
        fb.new_scenes(pts, poses_2d=p2d, drop_masks=drop)
        out = fb.run(iters, groups_per_graph=gpg).clone()
        for f in range(F):
            gm = model()
            hm0 = torch.zeros((V, J, H, W), device=dev)
            loop = MultiViewLoop(gm, cams, hm0, dataset=dataset, sparse=True, use_graph=use_graph, fused_tail=True)
            gm.reset_from_points(pts[f])
            for slots, vb, gt, stats, idx in loop.size_groups:
                generate_heatmaps(gm._xyz.detach(), gm.get_scaling.detach(), gm._rotation.detach(),
                                  torch.tensor(p2d[f][slots], device=dev), [cams[k] for k in slots], out=gt, views=vb,
                                  totals=stats.totals, drop_mask=drop[f][slots])
            loop.run(iters, groups_per_graph=gpg)
            assert torch.equal(out[f], gm._xyz.detach()), f"xyz of frame {f}"
            assert torch.equal(fb.scaling[f], gm._scaling.detach()) and torch.equal(fb.rotation[f], gm._rotation.detach()), f"frame {f}"
            assert torch.equal(fb.opacity[f], gm._opacity.detach()), f"opacity of frame {f}"
            assert torch.equal(fb.exp_avg[f], loop.exp_avg) and torch.equal(fb.exp_avg_sq[f], loop.exp_avg_sq), f"moments of frame {f}"
            assert torch.equal(fb.accumulated_grads[f], loop.accumulated_grads), f"slots of frame {f}"
            assert torch.equal(fb.counters[f], loop.counters), f"counters of frame {f}"
        moved = float((out - torch.tensor(pts, device=dev)).norm(dim=2).mean())
        return dict(moved_mm=moved, frames=F)
    except AssertionError as e:
        raise AssertionError(f"{tag} -> {str(e)[:300]}") from None


def run_dropin_case(seed, dev):
    """The drop-in surface: gaussian_renderer.render_* -> GaussianRasterizer -> autograd, with the model's LEAF parameters handed
    to the kernels (FAST_ACTIVATIONS: sigmoid / exp / normalize and their Jacobians in-kernel) against the literal path (torch
    activations around the call), on a random model: image, radii, and the gradients of every leaf."""
    import types
    import gaussian_renderer
    from gaussian_renderer import render_functions
    from skelsplat_amd.scene import SyntheticScene, GaussianModel, DATASETS
    from skelsplat_amd.loop import l2_loss_gaussian
    rng = np.random.default_rng(seed)
    dataset = str(rng.choice(["h36m", "panoptic", "occlusion-person"]))
    W, H = int(rng.integers(48, 260)), int(rng.integers(48, 200))
    aa = bool(rng.integers(0, 2))
    smod = float(rng.choice([1.0, 1.25, 0.8]))
    tag = f"drop-in seed {seed}: {dataset} {W}x{H} aa={aa} scaling_modifier={smod}"
    try:
        sc = SyntheticScene(dataset, n_views=2, seed=seed, W=W, H=H, ring=2500.0, fx=1145.0 * (W / 1000) * float(rng.uniform(0.8, 1.7)),
                            device=dev)
        res = []
        torch.manual_seed(seed)
        J = sc.n_joints
        d_scale = torch.randn(J, 3, device=dev) * 0.3
        d_rot = torch.randn(J, 4, device=dev) * 0.5
        d_op = torch.randn(J, 1, device=dev)
        gt = torch.rand(J, H, W, device=dev) * (torch.rand(J, H, W, device=dev) > 0.8)
        for fast in (True, False):
            gm = GaussianModel().create_from_points(sc.pose_3d_init, sc.spatial_lr_scale, J, scaling=float(3.0 + (seed % 7) * 0.2),
                                                    opacity_on=True, scene_type=dataset, device=dev)
            with torch.no_grad():
                gm._scaling += d_scale
                gm._rotation += d_rot
                gm._opacity.copy_(d_op)
            gm.training_setup()
            old = gaussian_renderer.FAST_ACTIVATIONS
            gaussian_renderer.FAST_ACTIVATIONS = fast
            try:
                pipe = types.SimpleNamespace(debug=False, antialiasing=aa, compute_cov3D_python=False, convert_SHs_python=False)
                pkg = render_functions[DATASETS[dataset]["rendering"]](sc.cameras[seed % 2], gm, pipe, torch.zeros(3, device=dev),
                                                                       scaling_modifier=smod)
                loss, _ = l2_loss_gaussian(pkg["render"], gt)
                loss.backward()
            finally:
                gaussian_renderer.FAST_ACTIVATIONS = old
            res.append(dict(render=pkg["render"].detach().cpu().numpy(), radii=pkg["radii"].cpu().numpy(),
                            xyz=gm._xyz.grad.cpu().numpy(), scaling=gm._scaling.grad.cpu().numpy(),
                            rotation=gm._rotation.grad.cpu().numpy(), opacity=gm._opacity.grad.cpu().numpy(),
                            screen=pkg["viewspace_points"].grad.cpu().numpy()))
        a, b = res
        assert np.array_equal(a["radii"], b["radii"]), "radii"
        util.assert_close("render", a["render"], b["render"], rtol=1e-5, atol_scale=1e-6)
        for k in ("xyz", "scaling", "rotation", "opacity", "screen"):
            util.assert_close(k, a[k], b[k], rtol=2e-4, atol_scale=1e-5)
        return dict(visible=float((a["radii"] > 0).sum()), grad=float(np.abs(b["xyz"]).max()))
    except AssertionError as e:
        raise AssertionError(f"{tag} -> {str(e)[:300]}") from None


def run_ops_case(seed, dev):
    """The smaller ops on random shapes against plain torch: fused masked L2 (any element count, 1-6 views), fused SSIM against the
    conv2d form (any image size), the 3-NN mean distance against brute force (both searches)."""
    import torch.nn.functional as Fn
    from skelsplat_amd import ops
    rng = np.random.default_rng(seed)
    g = torch.Generator(device=dev).manual_seed(seed)
    try:
        # masked L2
        V, C, H, W = int(rng.integers(1, 7)), int(rng.integers(1, 20)), int(rng.integers(1, 90)), int(rng.integers(1, 130))
        dens = float(rng.choice([0.0, 0.02, 0.5, 1.0]))
        r = torch.rand((V, C, H, W), device=dev, generator=g) * (torch.rand((V, C, H, W), device=dev, generator=g) < dens)
        t_ = torch.rand((V, C, H, W), device=dev, generator=g) * (torch.rand((V, C, H, W), device=dev, generator=g) < dens)
        dL, S, N = ops.masked_l2(r, t_)
        m = (t_ > 0) | (r > 0)
        assert torch.equal(N, m.sum(dim=(1, 2, 3)).double()), f"masked_l2 {V}x{C}x{H}x{W} density {dens}: N"
        wantS = (((r - t_).double() ** 2) * m).sum(dim=(1, 2, 3))
        assert ((S - wantS).abs() <= 1e-5 * wantS.abs() + 1e-12).all(), f"masked_l2 {V}x{C}x{H}x{W}: S {S} vs {wantS}"
        assert torch.equal(dL, 2.0 * (r - t_) * m), f"masked_l2 {V}x{C}x{H}x{W}: dL"
        # 3-NN mean squared distance
        P = int(rng.choice([1, 2, 3, 4, 5, 17, 19, 64, 300, 2500]))
        pts = torch.randn((P, 3), device=dev, generator=g) * float(rng.uniform(0.1, 1000.0))
        d2 = torch.cdist(pts.double(), pts.double()) ** 2
        d2.fill_diagonal_(float("inf"))
        k = min(3, P - 1)
        if P >= 4:
            want = d2.topk(3, largest=False).values.mean(dim=1).float()
            for method in ("allpairs", "grid"):
                got = ops.distCUDA2(pts, method=method)
                assert torch.allclose(got, want, rtol=2e-4, atol=1e-6 * float(want.max())), f"knn P={P} {method}"
        # fused SSIM
        B, Cs, Hs, Ws = int(rng.integers(1, 4)), int(rng.integers(1, 6)), int(rng.integers(1, 120)), int(rng.integers(1, 150))
        a = torch.rand((B, Cs, Hs, Ws), device=dev, generator=g).requires_grad_(True)
        b = torch.rand((B, Cs, Hs, Ws), device=dev, generator=g)
        from fused_ssim import fused_ssim
        got = fused_ssim(a, b)
        got.backward()
        ga = a.grad.clone()
        # the reference on the CPU in double (torch's conv2d BACKWARD through MIOpen on the GPU faults -- a GPU memory access fault --
        # for some of these shapes, e.g. (2, 1, 10, 20) and (2, 1, 88, 107): twice in 4 000 cases, in `ref.backward()`)
        a2 = a.detach().double().cpu().requires_grad_(True)
        b2 = b.double().cpu()
        win = torch.tensor([0.001028380123898387, 0.0075987582094967365, 0.036000773310661316, 0.10936068743467331,
                            0.21300552785396576, 0.26601171493530273, 0.21300552785396576, 0.10936068743467331,
                            0.036000773310661316, 0.0075987582094967365, 0.001028380123898387], dtype=torch.float64)
        w2 = (win[:, None] * win[None, :]).expand(Cs, 1, 11, 11).contiguous()
        conv = lambda x: Fn.conv2d(x, w2, padding=5, groups=Cs)
        mu1, mu2 = conv(a2), conv(b2)
        s1, s2, s12 = conv(a2 * a2) - mu1 * mu1, conv(b2 * b2) - mu2 * mu2, conv(a2 * b2) - mu1 * mu2
        ref = (((2 * mu1 * mu2 + 0.01 ** 2) * (2 * s12 + 0.03 ** 2)) / ((mu1 * mu1 + mu2 * mu2 + 0.01 ** 2) * (s1 + s2 + 0.03 ** 2))).mean()
        ref.backward()
        assert abs(got.item() - ref.item()) <= 2e-5 * abs(ref.item()) + 1e-6, f"ssim {B}x{Cs}x{Hs}x{Ws}: {got.item()} vs {ref.item()}"
        gref = a2.grad.float()
        assert torch.allclose(ga.cpu(), gref, rtol=2e-3, atol=2e-5 * float(gref.abs().max()) + 1e-9), f"ssim grad {B}x{Cs}x{Hs}x{Ws}"
        return dict(mask=float(N.sum()), points=float(P))
    except AssertionError as e:
        raise AssertionError(f"ops seed {seed} -> {str(e)[:300]}") from None
