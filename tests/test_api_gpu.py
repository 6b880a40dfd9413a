"""The Python half of the path held to what the REFERENCE's own Python does.

tests/golden/reference_api.npz was recorded by running /root/reference's gaussian_renderer.render_* ->
GaussianRasterizer -> _RasterizeGaussians over a recording `_C` (tests/golden/make_golden_api.py).  Here the
repository's render_* -> GaussianRasterizer -> forward_views / backward_views -> sks_forward / sks_backward runs on the
GPU with the same raw inputs, and every value that crosses into the native library is compared with what the reference
handed to ITS native module: argument by argument (gaussian_renderer/__init__.py:28-138,
DGR/diff_gaussian_rasterization_h36m/__init__.py:60-81, 101-139).
"""
import math
import os

import numpy as np
import pytest
import torch

from tests import util

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_api.npz")
KEYS = {"h36m": "diff-gaussian-rasterization-h36m", "panoptic": "diff-gaussian-rasterization-panoptic",
        "op": "diff-gaussian-rasterization-op"}


class _Pipe:
    convert_SHs_python = False
    compute_cov3D_python = False
    debug = False
    antialiasing = False


def _model_and_camera(G, key, dev):
    from skelsplat_amd.scene import GaussianModel, Camera
    pre = key + "_"
    ds = str(G[pre + "in_dataset"])
    W, H = [int(x) for x in G[pre + "in_WH"]]
    cam = Camera(1, G[pre + "in_cam_R"], G[pre + "in_cam_T"], G[pre + "in_cam_K"], W, H, device=dev)
    J = G[pre + "in_xyz"].shape[0]
    gm = GaussianModel().create_from_points(G[pre + "in_xyz"], 1.0, J, scaling=3.9, scaling_modifier=1.2, scene_type=ds, device=dev)
    with torch.no_grad():
        gm._scaling.copy_(torch.tensor(G[pre + "in_scaling"]))
        gm._rotation.copy_(torch.tensor(G[pre + "in_rotation"]))
        gm._opacity.copy_(torch.tensor(G[pre + "in_opacity"]))
    assert np.array_equal(gm._features_dc.detach().cpu().numpy(), G[pre + "in_features_dc"])       # (J,1,J) one-hot, gaussian_model.py:159-166
    assert gm.active_sh_degree == int(G[pre + "in_active_sh_degree"])
    pipe = type("Pipe", (_Pipe,), dict(compute_cov3D_python=bool(G[pre + "in_pipe"][0]), antialiasing=bool(G[pre + "in_pipe"][1])))
    return gm, cam, pipe, ds


@pytest.mark.gpu
@pytest.mark.parametrize("fast", [False, True], ids=["literal", "fast-activations"])
@pytest.mark.parametrize("key", ["h36m", "panoptic", "op"])
def test_render_hands_the_native_library_what_the_reference_hands_its_own(device, key, fast, monkeypatch):
    """`fast`: gaussian_renderer.FAST_ACTIVATIONS -- the model's LEAF parameters go to the rasterizer, which runs sigmoid / exp /
    normalize and their Jacobians in its kernels; held to the same reference run (activations of what was handed over ==
    the reference's activated arguments; images, loss and all five gradients to the same bars)."""
    import gaussian_renderer
    from gaussian_renderer import render_functions
    monkeypatch.setattr(gaussian_renderer, "FAST_ACTIVATIONS", fast)
    from skelsplat_amd import rasterizer as R, _lib
    from skelsplat_amd.loop import l2_loss_gaussian, limb_3d_consistency_loss
    G = np.load(GOLD)
    pre = key + "_"
    gm, cam, pipe, ds = _model_and_camera(G, key, device)
    # the camera the reference's scene/cameras.py:Camera builds from the same (R, T, K, W, H)
    assert np.allclose(cam.world_view_transform.cpu().numpy(), G[pre + "cam_world_view_transform"], rtol=0, atol=1e-6)
    assert np.allclose(cam.full_proj_transform.cpu().numpy(), G[pre + "cam_full_proj_transform"], rtol=1e-6, atol=1e-6)
    assert np.allclose(cam.camera_center.cpu().numpy(), G[pre + "cam_camera_center"], rtol=1e-6, atol=1e-3)
    assert np.allclose([cam.FoVx, cam.FoVy], G[pre + "cam_fov"], rtol=1e-12)

    rec = {}
    lib = _lib.load()
    real = dict(f=lib.sks_forward, b=lib.sks_backward, fv=R.forward_views, bv=R.backward_views)

    def fwd_views(views, *a, **k):
        rec["fv"] = (views, a, k)
        return real["fv"](views, *a, **k)

    def bwd_views(st, *a, **k):
        rec["bv"] = (st, tuple(x.clone() if torch.is_tensor(x) else x for x in a), k)
        return real["bv"](st, *a, **k)

    def sks_forward(*a):
        rec["f"] = a
        return real["f"](*a)

    def sks_backward(*a):
        rec["b"] = a
        return real["b"](*a)
    monkeypatch.setattr(R, "forward_views", fwd_views)
    monkeypatch.setattr(R, "backward_views", bwd_views)
    monkeypatch.setattr(lib, "sks_forward", sks_forward)
    monkeypatch.setattr(lib, "sks_backward", sks_backward)

    bg = torch.tensor(G[pre + "in_bg"], device=device)
    smod = float(G[pre + "in_scaling_modifier"])
    pkg = render_functions[KEYS[key]](cam, gm, pipe, bg, scaling_modifier=smod, use_trained_exp=False, separate_sh=False)
    assert sorted(pkg.keys()) == sorted(str(k) for k in G[pre + "out_keys"])
    image = pkg["render"]
    gt = torch.tensor(G[pre + "in_gt"], device=device)
    l2, _ = l2_loss_gaussian(image, gt)
    loss = l2 + limb_3d_consistency_loss(gm.get_xyz, ds) * 1e-5
    params = [gm.get_xyz, gm._scaling, gm._rotation, gm._opacity]
    grads = torch.autograd.grad(loss, params + [pkg["viewspace_points"]], create_graph=True, retain_graph=True)   # train.py:160-161

    # ---- forward: tensors handed to the batched entry point == the reference's 20-argument tuple ---------------------
    views, a, k = rec["fv"]
    means3D, feats, opac, scales, rots, cov = a[:6]
    P, C = G[pre + "fwd_means3D"].shape[0], G[pre + "fwd_sh"].shape[2]
    close = lambda got, want, name, rtol=2e-6: util.assert_close(name, got.detach().cpu().numpy().reshape(want.shape), want,
                                                                  rtol=rtol, atol_scale=1e-6)
    raw_call = fast and not bool(G[pre + "in_pipe"][0])      # (compute_cov3D_python keeps the literal call)
    assert bool(int(rec["f"][16]) & _lib.SKS_RAW_PARAMS) == raw_call
    opac_h, scales_h, rots_h = opac, scales, rots               # what was handed over (pointer checks below)
    if raw_call:      # the leaves: their activations are what the reference handed to its native module
        assert opac.data_ptr() == gm._opacity.data_ptr() and scales.data_ptr() == gm._scaling.data_ptr()
        opac, scales, rots = torch.sigmoid(opac), torch.exp(scales), torch.nn.functional.normalize(rots)
    close(means3D, G[pre + "fwd_means3D"], "means3D")
    close(opac, G[pre + "fwd_opacities"], "opacities")
    close(feats, G[pre + "fwd_sh"], "features (read from `sh`, quirk Q1)")
    assert G[pre + "fwd_colors_precomp"].size == 0                       # the reference passes an empty tensor, not None
    if bool(G[pre + "in_pipe"][0]):     # compute_cov3D_python: six numbers per Gaussian, no scales / rotations
        close(cov, G[pre + "fwd_cov3Ds_precomp"], "cov3D_precomp", rtol=1e-5)
        assert G[pre + "fwd_scales"].size == 0 and G[pre + "fwd_rotations"].size == 0
        assert scales is None or scales.numel() == 0
        assert rots is None or rots.numel() == 0
    else:
        close(scales, G[pre + "fwd_scales"], "scales")
        close(rots, G[pre + "fwd_rotations"], "rotations")
        assert G[pre + "fwd_cov3Ds_precomp"].size == 0 and (cov is None or cov.numel() == 0)
    close(views.viewmatrix, G[pre + "fwd_viewmatrix"], "viewmatrix")
    close(views.projmatrix, G[pre + "fwd_projmatrix"], "projmatrix")
    # ---- ... and what reached the C ABI (include/skelsplat_hip.h: sks_forward) ---------------------------------------
    f = rec["f"]
    assert tuple(f[:5]) == (1, P, C, int(G[pre + "fwd_image_width"]), int(G[pre + "fwd_image_height"]))
    assert f[5] == views.viewmatrix.data_ptr() and f[6] == views.projmatrix.data_ptr()
    assert np.float32(f[7][0]) == np.float32(G[pre + "fwd_tanfovx"]) and np.float32(f[8][0]) == np.float32(G[pre + "fwd_tanfovy"])
    ptr = lambda t: None if t is None or t.numel() == 0 else t.data_ptr()
    assert f[9] == means3D.data_ptr() and f[11] == opac_h.data_ptr()
    assert f[10] == feats.reshape(P, -1).data_ptr()
    assert (f[12], f[13], f[14]) == (ptr(scales_h), ptr(rots_h), ptr(cov))   # NULL == the reference's empty-tensor sentinel (Q10)
    assert f[15] == pytest.approx(float(G[pre + "fwd_scale_modifier"])) and float(G[pre + "fwd_scale_modifier"]) == smod
    flags = int(f[16])
    assert bool(flags & _lib.SKS_ANTIALIASING) == bool(G[pre + "fwd_antialiasing"])
    assert bool(flags & _lib.SKS_DEBUG_SYNC) == bool(G[pre + "fwd_debug"])
    assert flags & _lib.SKS_CLAMP01                                      # gaussian_renderer/__init__.py:129 folded into the store
    assert not bool(G[pre + "fwd_prefiltered"]) and int(G[pre + "fwd_sh_degree"]) == gm.active_sh_degree
    # ---- the render package ------------------------------------------------------------------------------------------
    assert np.array_equal(pkg["radii"].cpu().numpy(), G[pre + "out_radii"]) and pkg["radii"].dtype == torch.int32
    assert np.array_equal(pkg["visibility_filter"].cpu().numpy(), G[pre + "out_visibility_filter"])
    assert pkg["visibility_filter"].dtype == torch.int64
    util.assert_close("render", image.detach().cpu().numpy(), G[pre + "out_render"], rtol=1e-4, atol_scale=1e-5)
    util.assert_close("depth", pkg["depth"].detach().cpu().numpy(), G[pre + "out_depth"], rtol=1e-4, atol_scale=1e-5)
    assert np.array_equal(pkg["viewspace_points"].detach().cpu().numpy(), G[pre + "out_viewspace_points"])
    assert abs(loss.item() - float(G[pre + "loss"])) <= 1e-5 * abs(float(G[pre + "loss"]))
    # ---- backward: the 24-argument tuple -----------------------------------------------------------------------------
    st, ba, bk = rec["bv"]
    dL_color, dL_inv = ba[6], (ba[7] if len(ba) > 7 else bk.get("dL_dinvdepth"))
    # the reference's autograd hands `_C` the gradient w.r.t. the UNclamped image (clamp's backward already applied);
    # here the clamp lives in the kernels: gradient w.r.t. the clamped image x the pass-through mask is the same thing
    raw = torch.tensor(G[pre + "out_render"], device=device)
    want = G[pre + "bwd_grad_out_color"]
    got = dL_color.detach().reshape(want.shape)
    inside = raw < 1
    util.assert_close("grad_out_color", (got * inside).cpu().numpy(), want * inside.cpu().numpy(), rtol=1e-4, atol_scale=1e-6)
    assert not G[pre + "bwd_grad_out_depth"].any() and (dL_inv is None or not bool(dL_inv.any()))   # Q4: materialised zeros
    b = rec["b"]
    assert tuple(b[:5]) == tuple(f[:5]) and b[16] == pytest.approx(smod)
    assert np.array_equal(G[pre + "bwd_radii"], G[pre + "out_radii"]) and int(G[pre + "bwd_num_rendered"]) >= 0
    bg_np = G[pre + "bwd_bg"]
    assert (b[9] is None) == (not bg_np.any())                           # an all-zero background selects the bg-free kernels
    # ---- gradients on the four leaves + the screen-space points (the 9-slot return, __init__.py:129-139) ---------------
    for name, g in zip(("xyz", "scaling", "rotation", "opacity", "viewspace_points"), grads):
        util.assert_close("grad " + name, g.detach().cpu().numpy(), G[pre + "grad_" + name], rtol=2e-3, atol_scale=2e-4)
    assert float(np.abs(G[pre + "grad_viewspace_points"][:, :2]).max()) > 0 and not G[pre + "grad_viewspace_points"][:, 2].any()


@pytest.mark.gpu
def test_gradient_slots_and_mark_visible_like_the_reference(device):
    """Which native gradient lands on which of the 9 inputs of rasterize_gaussians (the reference, run over a marker `_C`:
    means3D <- dL_dmeans3D, means2D <- dL_dmeans2D, colors_precomp <- dL_dcolors, opacities, scales, rotations,
    cov3Ds_precomp; sh <- dL_dsh, which the reference's kernel fills with garbage (Q5) and this library with the true
    dL/dfeature), and markVisible on the golden points."""
    from skelsplat_amd import rasterizer as R
    import diff_gaussian_rasterization_h36m as dgr
    G = np.load(GOLD)
    assert G["slot_order"].tolist() == ["means3D", "means2D", "sh", "colors_precomp", "opacities", "scales", "rotations", "cov3Ds_precomp"]
    # marker k+1 = position k of the native 8-tuple (rasterize_points.cu:222)
    native = ["dL_dmeans2D", "dL_dcolors", "dL_dopacity", "dL_dmeans3D", "dL_dcov3D", "dL_dsh", "dL_dscales", "dL_drotations"]
    landed = {str(n): native[int(m) - 1] for n, m in zip(G["slot_order"], G["slot_marker"])}
    assert landed == dict(means3D="dL_dmeans3D", means2D="dL_dmeans2D", sh="dL_dsh", colors_precomp="dL_dcolors",
                          opacities="dL_dopacity", scales="dL_dscales", rotations="dL_drotations", cov3Ds_precomp="dL_dcov3D")
    # the same routing here: every input's .grad equals the raw batched backward's entry of that name
    c = util.make_case(seed=7, W=96, H=80, scale_log=4.0, n_views=1)
    t = lambda a: torch.tensor(a, device=device)
    cam = c.cams[0].to(device)
    import math
    rs = dgr.GaussianRasterizationSettings(c.H, c.W, math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5), torch.zeros(3, device=device),
                                           1.0, cam.world_view_transform, cam.full_proj_transform, 0, cam.camera_center, False, False, False)
    ins = dict(means3D=t(c.means), means2D=torch.zeros(c.P, 3, device=device), sh=t(c.feat)[:, None, :].contiguous(),
               opacities=t(c.opac), scales=t(c.scales), rotations=t(c.quats))
    for x in ins.values():
        x.requires_grad_(True)
    color, radii, inv = dgr.GaussianRasterizer(rs)(means3D=ins["means3D"], means2D=ins["means2D"], opacities=ins["opacities"],
                                                   shs=ins["sh"], scales=ins["scales"], rotations=ins["rotations"])
    dLc, dLi = t(c.dL_color[0]), t(c.dL_inv[0])
    ((color * dLc).sum() + (inv * dLi).sum()).backward()
    views = R.ViewBatch.from_settings(rs)
    args = (ins["means3D"].detach(), t(c.feat), ins["opacities"].detach(), ins["scales"].detach(), ins["rotations"].detach(), None)
    st = R.forward_views(views, *args)[3]
    raw = R.backward_views(st, *args, dLc[None], dLi[None], want_dfeatures=True)
    for name, key in (("means3D", "means3D"), ("means2D", "means2D"), ("opacities", "opacities"), ("scales", "scales"),
                      ("rotations", "rotations"), ("sh", "features")):
        assert torch.equal(ins[name].grad.reshape(raw[key][0].shape), raw[key][0]), name
    # markVisible (rasterizer_impl.cu:54-66) on the reference-run points
    rs2 = rs._replace(viewmatrix=t(G["h36m_cam_world_view_transform"]), projmatrix=t(G["h36m_cam_full_proj_transform"]))
    got = dgr.GaussianRasterizer(rs2).markVisible(t(G["mark_points"]))
    assert got.dtype == torch.bool and np.array_equal(got.cpu().numpy(), G["mark_visible"])


def test_settings_fields_and_validation_messages_like_the_reference():
    """CPU: the NamedTuple's fields and GaussianRasterizer.forward's argument validation (DGR __init__.py:143-156, 178-182)
    against what the reference's own class raised for the same five calls."""
    import diff_gaussian_rasterization_h36m as dgr
    G = np.load(GOLD)
    assert list(dgr.GaussianRasterizationSettings._fields) == G["settings_fields"].tolist()
    P, C = 5, 17
    rs = dgr.GaussianRasterizationSettings(32, 48, 0.5, 0.4, torch.zeros(3), 1.0, torch.eye(4), torch.eye(4), 0, torch.zeros(3),
                                           False, False, False)
    rast = dgr.GaussianRasterizer(rs)
    m3, m2, op = torch.zeros(P, 3), torch.zeros(P, 3), torch.ones(P, 1)
    shs, sc, rot = torch.ones(P, 1, C), torch.ones(P, 3), torch.ones(P, 4)
    calls = (dict(), dict(shs=shs, colors_precomp=torch.ones(P, C)), dict(shs=shs), dict(shs=shs, scales=sc),
             dict(shs=shs, scales=sc, rotations=rot, cov3D_precomp=torch.ones(P, 6)))
    for kw, want in zip(calls, G["validation_messages"].tolist()):
        with pytest.raises(Exception) as e:
            rast(means3D=m3, means2D=m2, opacities=op, **kw)
        assert f"{type(e.value).__name__}: {e.value}" == want


def test_reference_api_fixture_is_self_consistent():
    """CPU: the recorded tuples have the reference's arity, order and dtypes; what it hands to `_C` is what its
    GaussianModel activations produce from the stored raw parameters (gaussian_model.py:32-47)."""
    G = np.load(GOLD)
    assert len(G["fwd_names"]) == 20 and len(G["bwd_names"]) == 24
    for key in G["scenarios"].tolist():
        pre = key + "_"
        xyz, s, q, o = (torch.tensor(G[pre + "in_" + n]) for n in ("xyz", "scaling", "rotation", "opacity"))
        assert np.array_equal(G[pre + "fwd_means3D"], xyz.numpy())
        assert np.allclose(G[pre + "fwd_opacities"], torch.sigmoid(o).numpy(), rtol=1e-6)
        if G[pre + "fwd_scales"].size:
            assert np.allclose(G[pre + "fwd_scales"], torch.exp(s).numpy(), rtol=1e-6)
            assert np.allclose(G[pre + "fwd_rotations"], torch.nn.functional.normalize(q).numpy(), rtol=1e-6, atol=1e-7)
        else:
            assert G[pre + "fwd_cov3Ds_precomp"].shape == (xyz.shape[0], 6)
        assert G[pre + "fwd_sh"].shape == (xyz.shape[0], 1, xyz.shape[0]) and str(G[pre + "fwd_sh__dtype"]) == "torch.float32"
        assert str(G[pre + "fwd_image_height__dtype"]) == "int" and str(G[pre + "fwd_tanfovx__dtype"]) == "float"
        assert G[pre + "bwd_grad_out_color"].shape == G[pre + "out_render"].shape
        assert G[pre + "bwd_grad_out_depth"].shape == G[pre + "out_depth"].shape
        for n in ("means3D", "opacities", "scales", "rotations", "cov3Ds_precomp", "sh", "viewmatrix", "projmatrix", "bg", "campos"):
            assert np.array_equal(G[pre + "fwd_" + n], G[pre + "bwd_" + n]), n


def _integration_md_binding(num_channels, namespace=False):
    """The `_C` class of INTEGRATION.md section B, extracted from the document and executed as written (library path and
    NUM_CHANNELS substituted)."""
    import re
    from skelsplat_amd import _lib
    _lib.load()      # builds / finds the library; the stub below opens the same file through its own ctypes handle
    text = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "INTEGRATION.md")).read()
    blocks = re.findall(r"```python\n(.*?)```", text, flags=re.S)
    code = [b for b in blocks if "class _C" in b and "sks_forward" in b]
    assert len(code) == 1, "INTEGRATION.md section B must hold exactly one binding block"
    src = code[0].replace('ctypes.CDLL("libskelsplat_hip.so")', f'ctypes.CDLL({_lib.LIB_PATH!r})')
    src = re.sub(r"NUM_CHANNELS = 17\b", f"NUM_CHANNELS = {num_channels}", src, count=1)
    ns = {}
    exec(compile(src, "INTEGRATION.md#B", "exec"), ns)
    return ns if namespace else ns["_C"]


@pytest.mark.gpu
@pytest.mark.parametrize("key", ["h36m", "panoptic", "op"])
def test_integration_md_binding_reproduces_the_reference_run(device, key):
    """The ctypes binding INTEGRATION.md shows a maintainer, fed with the 20- and 24-argument tuples the reference's own
    Python passed to its native module, returns what the reference run received from it (7-tuple / 8-tuple layouts of
    rasterize_points.cu:124, 222): colour planes, radii, inverse depth, and every gradient but dL_dsh (quirk Q5)."""
    G = np.load(GOLD)
    pre = key + "_"
    C = G[pre + "fwd_sh"].shape[2]
    stub = _integration_md_binding(C)

    def arg(kind, name):
        a, dt = G[f"{pre}{kind}_{name}"], str(G[f"{pre}{kind}_{name}__dtype"])
        if dt.startswith("torch."):
            t = torch.tensor(a)
            return t.to(device) if t.numel() else t              # empty CPU tensors are the "not provided" sentinel (Q10)
        return {"int": int, "float": float, "bool": bool}[dt](a)
    fwd = [arg("fwd", str(n)) for n in G["fwd_names"]]
    num_rendered, color, radii, geom, binning, img, invd = stub.rasterize_gaussians(*fwd)
    assert color.shape == G[pre + "ret_color"].shape and radii.dtype == torch.int32
    assert np.array_equal(radii.cpu().numpy(), G[pre + "ret_radii"])
    assert np.array_equal(color.cpu().numpy(), G[pre + "ret_color"])              # same fp32 arithmetic: bit for bit
    assert np.array_equal(invd.cpu().numpy(), G[pre + "ret_invdepth"])
    bwd = [arg("bwd", str(n)) for n in G["bwd_names"]]
    names = [str(n) for n in G["bwd_names"]]
    bwd[names.index("radii")], bwd[names.index("geomBuffer")] = radii, geom
    bwd[names.index("binningBuffer")], bwd[names.index("imgBuffer")], bwd[names.index("num_rendered")] = binning, img, num_rendered
    out = stub.rasterize_gaussians_backward(*bwd)
    order = ("dL_dmeans2D", "dL_dcolors", "dL_dopacity", "dL_dmeans3D", "dL_dcov3D", "dL_dsh", "dL_dscales", "dL_drotations")
    assert len(out) == 8
    for name, got in zip(order, out):
        want = G[pre + "ret_" + name]
        assert tuple(got.shape) == want.shape, name
        if name == "dL_dsh":
            util.assert_close(name + " (= dL_dcolors here)", got.cpu().numpy().reshape(-1), G[pre + "ret_dL_dcolors"].reshape(-1))
        elif want.any() or name not in ("dL_dscales", "dL_drotations"):
            util.assert_close(name, got.cpu().numpy(), want)
    pts = torch.tensor(G["mark_points"], device=device)
    vm, pm = torch.tensor(G["h36m_cam_world_view_transform"], device=device), torch.tensor(G["h36m_cam_full_proj_transform"], device=device)
    assert np.array_equal(stub.mark_visible(pts, vm, pm).cpu().numpy(), G["mark_visible"])


@pytest.mark.gpu
def test_integration_md_binding_carries_the_stress_scene(device):
    """P > 256 through the reference's call shape (rasterize_points.h:18-71): the INTEGRATION.md binding owns the binning arena
    like the reference owns its binningBuffer (rasterize_points.cu:80-85), reads num_rendered back like rasterizer_impl.cu:283-288
    and grows an arena that was too small.  One 2048 x 2048 view of BASELINE's stress scene (P = 4 352): num_rendered equals the
    oracle's pair count, images and gradients equal the repository's own surface bit for bit -- from a first arena of 64 pairs."""
    from skelsplat_amd.scene import stress_scene
    from skelsplat_amd import rasterizer as R
    from oracle import oracle as orc
    ns = _integration_md_binding(17, namespace=True)
    stub = ns["_C"]
    W = H = 2048
    sc, g = stress_scene(1, W=W, H=H)
    cam = sc.cameras[0].to(device)
    tt = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
    means, feat, opac, scales, quats = (tt(g[k]) for k in ("means", "feat", "opac", "scales", "quats"))
    P = means.shape[0]
    assert P == 4352
    tfx, tfy = math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5)
    ocam = orc.Cam(W, H, tfx, tfy, cam.world_view_transform.cpu().numpy(), cam.full_proj_transform.cpu().numpy())
    want_R = int(orc.preprocess(g["means"], g["opac"], g["scales"], g["quats"], None, ocam)["tiles_touched"].astype(np.int64).sum())
    ns["_ARENA"][(P, W, H)] = 64                      # far too small: the binding must grow it and render again
    empty = torch.Tensor([])
    bg = torch.zeros(3, device=device)
    sh = feat.reshape(P, 1, 17)
    fwd = (bg, means, empty, opac, scales, quats, 1.0, empty, cam.world_view_transform, cam.full_proj_transform, tfx, tfy, H, W,
           sh, 0, cam.camera_center, False, False, False)
    num_rendered, color, radii, geom, binning, img, invd = stub.rasterize_gaussians(*fwd)
    assert num_rendered == want_R and ns["_ARENA"][(P, W, H)] >= want_R
    views = R.ViewBatch.from_cameras([cam])
    col, inv, rad, st = R.forward_views(views, means, feat, opac, scales, quats, None)
    assert torch.equal(color, col[0]) and torch.equal(invd, inv[0]) and torch.equal(radii, rad[0])
    gen = torch.Generator(device=device).manual_seed(3)
    dL = torch.randn((17, H, W), device=device, generator=gen)
    dLi = torch.randn((1, H, W), device=device, generator=gen)
    out = stub.rasterize_gaussians_backward(bg, means, radii, empty, opac, scales, quats, 1.0, empty, cam.world_view_transform,
                                            cam.full_proj_transform, tfx, tfy, dL, dLi, sh, 0, cam.camera_center, geom, num_rendered,
                                            binning, img, False, False)
    gr = R.backward_views(st, means, feat, opac, scales, quats, None, dL[None], dLi[None], want_dfeatures=True)
    order = ("means2D", "features", "opacities", "means3D", "cov3D", None, "scales", "rotations")
    for name, got in zip(order, out):
        if name is not None and name != "features":
            assert torch.equal(got, gr[name][0]), name
    # (dL/dfeatures is accumulated with float atomics on the binned path: same sum, any order)
    util.assert_close("dL_dcolors", out[1].cpu().numpy(), gr["features"][0].cpu().numpy(), rtol=1e-4, atol_scale=1e-5)
    assert torch.isfinite(out[3]).all() and float(out[3].abs().max()) > 0
    # a second call of the shape finds the grown arena: one forward, no retry
    n2, color2, *_ = stub.rasterize_gaussians(*fwd)
    assert n2 == want_R and torch.equal(color2, color)
