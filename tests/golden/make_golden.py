"""Generates tests/golden/reference_python.npz by importing the REFERENCE's own Python (device-agnostic parts only)
from /root/reference in the build container.  The reference cannot travel to the GPU box, so only these small
input/output vectors are committed.  Run:  python tests/golden/make_golden.py

Pins: camera matrices (utils/graphics_utils.py), LR schedule (utils/general_utils.py:38-71),
masked-L2 loss value+gradient (utils/loss_utils.py:86-100), limb-symmetry loss value+gradient (:226-250),
SSIM value+gradient (:253-300 -- the same function fused-ssim's own test uses as oracle),
3D covariance from scaling + rotation (utils/general_utils.py:61-119, the Python twin of computeCov3D),
scene.cameras.Camera matrices and scene.gaussian_model.GaussianModel initialisation / optimiser groups.
"""
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def main():
    for n in ("tensordict", "cupy", "cupyx", "cupyx.scipy", "cupyx.scipy.ndimage", "plyfile", "cv2"):
        _stub(n)
    sys.modules["tensordict"].TensorDict = dict
    sys.modules["cupyx.scipy.ndimage"].gaussian_filter = None
    sys.modules["plyfile"].PlyData = sys.modules["plyfile"].PlyElement = None
    sys.path.insert(0, REF)
    from utils import graphics_utils, loss_utils, general_utils

    out = {}
    rng = np.random.default_rng(0)
    g = torch.Generator().manual_seed(0)

    # cameras: random rotations / translations / intrinsics
    Rs, Ts, Ks, W2V, PROJ, FOV = [], [], [], [], [], []
    for i in range(6):
        A = rng.normal(size=(3, 3))
        Q, _ = np.linalg.qr(A)
        if np.linalg.det(Q) < 0:
            Q[:, 0] *= -1
        T = rng.normal(scale=2000.0, size=3)
        W, H = [(1000, 1000), (1002, 1000), (1920, 1080), (1280, 720), (160, 128), (2048, 2048)][i]
        K = np.array([[1145.0 + 10 * i, 0, W / 2 + rng.uniform(-15, 15)], [0, 1143.0 + 7 * i, H / 2 + rng.uniform(-15, 15)], [0, 0, 1]])
        Rs.append(Q); Ts.append(T); Ks.append(K)
        W2V.append(graphics_utils.getWorld2View2(Q, T))
        PROJ.append(graphics_utils.getProjectionMatrix2(0.01, 100.0, K, W, H).numpy())
        FOV.append([graphics_utils.focal2fov(K[0, 0], W), graphics_utils.focal2fov(K[1, 1], H), W, H])
    out.update(cam_R=np.stack(Rs), cam_T=np.stack(Ts), cam_K=np.stack(Ks), cam_w2v=np.stack(W2V),
               cam_proj=np.stack(PROJ), cam_fov=np.array(FOV))

    # LR schedule (configs/h36m.yaml:53-57 values, spatial_lr_scale 5500)
    f = general_utils.get_expon_lr_func(lr_init=0.0005 * 5500.0, lr_final=0.000005 * 5500.0, lr_delay_mult=0.0, max_steps=4000)
    out["lr_steps"] = np.arange(0, 501)
    out["lr_values"] = np.array([f(int(s)) for s in out["lr_steps"]], dtype=np.float64)
    f2 = general_utils.get_expon_lr_func(0.01, 0.001, lr_delay_steps=100, lr_delay_mult=0.1, max_steps=500)
    out["lr2_values"] = np.array([f2(int(s)) for s in out["lr_steps"]], dtype=np.float64)

    # masked L2 (value + autograd gradient)
    r = (torch.rand((5, 24, 20), generator=g) * (torch.rand((5, 24, 20), generator=g) > 0.6)).requires_grad_(True)
    t = torch.rand((5, 24, 20), generator=g) * (torch.rand((5, 24, 20), generator=g) > 0.5)
    loss, err = loss_utils.l2_loss_gaussian(r, t, None)
    loss.backward()
    out.update(l2_render=r.detach().numpy(), l2_gt=t.numpy(), l2_loss=np.float64(loss.item()), l2_grad=r.grad.numpy(),
               l2_error=err.detach().numpy())

    # limb-symmetry loss for the three dataset conventions
    for key, root, J in (("h36m", "data/h36m", 17), ("panoptic", "data/panoptic", 19), ("occlusion-person", "data/occlusion-person", 15)):
        x = (torch.randn((J, 3), generator=g) * 300).requires_grad_(True)
        l = loss_utils.limb_3d_consistency_loss(x, root)
        l.backward()
        out[f"limb_{key}_xyz"] = x.detach().numpy()
        out[f"limb_{key}_loss"] = np.float64(l.item())
        out[f"limb_{key}_grad"] = x.grad.numpy()

    # SSIM (value per element via size_average + gradient)
    a = torch.rand((2, 3, 40, 52), generator=g).requires_grad_(True)
    b = torch.rand((2, 3, 40, 52), generator=g)
    s = loss_utils.ssim(a, b)
    s.backward()
    out.update(ssim_img1=a.detach().numpy(), ssim_img2=b.numpy(), ssim_value=np.float64(s.item()), ssim_grad=a.grad.numpy())

    # 3D covariance of the Gaussians, the Python twin of computeCov3D (scene/gaussian_model.py:33-37:
    # strip_symmetric(L @ L^T) with L = build_scaling_rotation(modifier * scaling, rotation), general_utils.py:61-119).
    # Those helpers hard-code device="cuda" in their torch.zeros calls; torch.zeros is wrapped for the duration of the
    # call so that the reference's own code runs on the CPU of the build container.
    cs = torch.exp(torch.randn((12, 3), generator=g) * 0.4 + 3.0)
    cq = torch.randn((12, 4), generator=g)
    real_zeros = torch.zeros

    def zeros_cpu(*a, **k):
        k.pop("device", None)
        return real_zeros(*a, **k)

    torch.zeros = zeros_cpu
    try:
        for tag, mod in (("", 1.0), ("_mod", 0.7)):
            L = general_utils.build_scaling_rotation(mod * cs, cq)
            out["cov_six" + tag] = general_utils.strip_symmetric(L @ L.transpose(1, 2)).numpy()
        out["cov_R"] = general_utils.build_rotation(cq).numpy()
    finally:
        torch.zeros = real_zeros
    out.update(cov_scaling=cs.numpy(), cov_rotation=cq.numpy())

    # scene.cameras.Camera and scene.gaussian_model.GaussianModel are hard-wired to device "cuda" (.cuda() calls and
    # device="cuda" keywords).  For the duration of the calls Tensor.cuda is the identity and the factory functions drop
    # the keyword, so that the reference's own constructors run on the CPU of the build container.
    real = {k: getattr(torch, k) for k in ("zeros", "ones", "tensor", "eye")}
    real_cuda = torch.Tensor.cuda

    def strip(fn):
        def wrapped(*a, **k):
            k.pop("device", None)
            return fn(*a, **k)
        return wrapped

    for k, fn in real.items():
        setattr(torch, k, strip(fn))
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        from scene.cameras import Camera
        from scene.gaussian_model import GaussianModel
        from utils.graphics_utils import BasicPointCloud
        wv, pj, fp, cc = [], [], [], []
        for i in range(6):
            W, H = int(out["cam_fov"][i, 2]), int(out["cam_fov"][i, 3])
            cam = Camera((W, H), i, out["cam_R"][i], out["cam_T"][i], out["cam_fov"][i, 0], out["cam_fov"][i, 1], out["cam_K"][i],
                         None, None, None, f"cam{i}", i, data_device="cpu")
            wv.append(cam.world_view_transform.numpy()); pj.append(cam.projection_matrix.numpy())
            fp.append(cam.full_proj_transform.numpy()); cc.append(cam.camera_center.numpy())
        out.update(camobj_world_view=np.stack(wv), camobj_projection=np.stack(pj), camobj_full_proj=np.stack(fp),
                   camobj_center=np.stack(cc))

        targs = types.SimpleNamespace(percent_dense=0.01, position_lr_init=0.0005, position_lr_final=0.000005,
                                      position_lr_delay_mult=0.0, position_lr_max_steps=4000, feature_lr=0.0, opacity_lr=0.0,
                                      scaling_lr=0.005, rotation_lr=0.001, exposure_lr_init=0.01, exposure_lr_final=0.001,
                                      exposure_lr_delay_steps=0, exposure_lr_delay_mult=0.0, iterations=500)
        for key, J in (("h36m", 17), ("panoptic", 19), ("occlusion-person", 15)):
            pts = rng.normal(scale=400.0, size=(J, 3))
            pcd = BasicPointCloud(points=pts, colors=np.zeros((J, 3)), normals=np.zeros((J, 3)))
            gmod = GaussianModel(1)
            gmod.create_from_pcd(pcd, [types.SimpleNamespace(image_name="a")], 5500.0, True, 3.0, J, 1.5, key)
            gmod.training_setup(targs)
            pre = f"gm_{key}_"
            out[pre + "points"] = pts
            out[pre + "xyz"] = gmod._xyz.detach().numpy()
            out[pre + "features_dc"] = gmod._features_dc.detach().numpy()
            out[pre + "scaling"] = gmod._scaling.detach().numpy()
            out[pre + "rotation"] = gmod._rotation.detach().numpy()
            out[pre + "opacity"] = gmod._opacity.detach().numpy()
            out[pre + "get_scaling"] = gmod.get_scaling.detach().numpy()
            out[pre + "get_opacity"] = gmod.get_opacity.detach().numpy()
            out[pre + "get_rotation"] = gmod.get_rotation.detach().numpy()
            out[pre + "group_names"] = np.array([g_["name"] for g_ in gmod.optimizer.param_groups])
            out[pre + "group_lrs"] = np.array([g_["lr"] for g_ in gmod.optimizer.param_groups], dtype=np.float64)
            out[pre + "adam_eps_betas"] = np.array([gmod.optimizer.defaults["eps"], *gmod.optimizer.defaults["betas"]], dtype=np.float64)
            out[pre + "xyz_lr_at"] = np.array([gmod.update_learning_rate(it) for it in (1, 4, 100, 500)], dtype=np.float64)
    finally:
        for k, fn in real.items():
            setattr(torch, k, fn)
        torch.Tensor.cuda = real_cuda

    np.savez_compressed(os.path.join(HERE, "reference_python.npz"), **out)
    print("wrote", os.path.join(HERE, "reference_python.npz"), {k: np.asarray(v).shape for k, v in out.items()})


if __name__ == "__main__":
    main()
