"""Host-side scene pieces that feed the hot path: cameras, skeleton Gaussians, synthetic scenes.

These are the inputs of SURVEY.md §8a (the reference builds them in scene/cameras.py, utils/graphics_utils.py,
scene/gaussian_model.py and scene/dataset_readers.py).  Only what the rasterizer / multi-view loop consume is
restated; dataset walkers, ply IO and densification are out of scope.
"""
import math

import numpy as np
import torch

# ------------------------------------------------------------------------------------------------------------
# camera math  (reference: utils/graphics_utils.py:38-49, 74-95, 101-102; scene/cameras.py:88-100)
# ------------------------------------------------------------------------------------------------------------


def focal2fov(focal, pixels):
    return 2 * math.atan(pixels / (2 * focal))


def fov2focal(fov, pixels):
    return pixels / (2 * math.tan(fov / 2))


def world2view2(R, t, translate=np.array([0.0, 0.0, 0.0]), scale=1.0):
    """utils/graphics_utils.py:38-49.  R is the camera-to-world rotation (the reference stores R transposed)."""
    Rt = np.zeros((4, 4))
    Rt[:3, :3] = R.transpose()
    Rt[:3, 3] = t
    Rt[3, 3] = 1.0
    C2W = np.linalg.inv(Rt)
    cam_center = C2W[:3, 3]
    cam_center = (cam_center + translate) * scale
    C2W[:3, 3] = cam_center
    Rt = np.linalg.inv(C2W)
    return np.float32(Rt)


def projection_matrix2(znear, zfar, K, W, H):
    """utils/graphics_utils.py:74-95 (off-centre projection from intrinsics)."""
    fx, fy, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    top = znear * cy / fy
    bottom = -znear * (H - cy) / fy
    right = znear * (W - cx) / fx
    left = -znear * cx / fx
    P = torch.zeros(4, 4)
    z_sign = 1.0
    P[0, 0] = 2.0 * znear / (right - left)
    P[1, 1] = 2.0 * znear / (top - bottom)
    P[0, 2] = -(right + left) / (right - left)
    P[1, 2] = (top + bottom) / (top - bottom)
    P[3, 2] = z_sign
    P[2, 2] = z_sign * zfar / (zfar - znear)
    P[2, 3] = -(zfar * znear) / (zfar - znear)
    return P


class Camera:
    """Attribute-compatible subset of scene/cameras.py:Camera used by render_* and the loop."""

    def __init__(self, uid, R, T, K, width, height, device="cpu", znear=0.01, zfar=100.0):
        self.uid = uid
        self.R = np.asarray(R, dtype=np.float64)
        self.T = np.asarray(T, dtype=np.float64)
        self.K = np.asarray(K, dtype=np.float64)
        self.image_width = int(width)
        self.image_height = int(height)
        self.FoVx = focal2fov(self.K[0, 0], width)
        self.FoVy = focal2fov(self.K[1, 1], height)
        self.znear, self.zfar = znear, zfar
        wvt = torch.tensor(world2view2(self.R, self.T)).transpose(0, 1)
        proj = projection_matrix2(znear, zfar, self.K, self.image_width, self.image_height).transpose(0, 1)
        full = wvt.unsqueeze(0).bmm(proj.unsqueeze(0)).squeeze(0)
        center = wvt.inverse()[3, :3]
        self.world_view_transform = wvt.contiguous().to(device)
        self.projection_matrix = proj.contiguous().to(device)
        self.full_proj_transform = full.contiguous().to(device)
        self.camera_center = center.contiguous().to(device)

    def to(self, device):
        for k in ("world_view_transform", "projection_matrix", "full_proj_transform", "camera_center"):
            setattr(self, k, getattr(self, k).to(device))
        return self


def look_at_camera(uid, position, target, fx, fy, cx, cy, width, height, device="cpu"):
    """OpenCV-style camera (x right, y down, z forward) at `position` looking at `target`, world z up."""
    position = np.asarray(position, dtype=np.float64)
    fwd = np.asarray(target, dtype=np.float64) - position
    fwd /= np.linalg.norm(fwd)
    right = np.cross(fwd, np.array([0.0, 0.0, 1.0]))
    right /= np.linalg.norm(right)
    down = np.cross(fwd, right)
    R_w2c = np.stack([right, down, fwd], 0)
    t = -R_w2c @ position
    K = np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1.0]])
    return Camera(uid, R_w2c.T, t, K, width, height, device=device)


def cameras_extent(cams):
    """scene/dataset_readers.py:482-503 (getNerfppNorm radius): 1.1 * max distance of a camera centre to their mean."""
    centers = []
    for c in cams:
        W2C = world2view2(c.R, c.T)
        centers.append(np.linalg.inv(W2C)[:3, 3:4])
    centers = np.hstack(centers)
    avg = np.mean(centers, axis=1, keepdims=True)
    return float(np.max(np.linalg.norm(centers - avg, axis=0, keepdims=True)) * 1.1)


# ------------------------------------------------------------------------------------------------------------
# skeleton templates (mm, z up, pelvis at z=900) and dataset conventions
# ------------------------------------------------------------------------------------------------------------

_H36M = [  # 0 pelvis 1-3 R leg 4-6 L leg 7 spine 8 thorax 9 neck 10 head 11-13 L arm 14-16 R arm
    (0, 0, 900), (-130, 0, 900), (-130, 0, 480), (-130, 0, 60), (130, 0, 900), (130, 0, 480), (130, 0, 60),
    (0, 0, 1150), (0, 0, 1400), (0, 0, 1500), (0, 0, 1650), (180, 0, 1400), (460, 0, 1400), (720, 0, 1400),
    (-180, 0, 1400), (-460, 0, 1400), (-720, 0, 1400)]
_PANOPTIC = [  # 0 neck 1 nose 2 body centre 3-5 L arm 6-8 L leg 9-11 R arm 12-14 R leg 15-18 eyes/ears
    (0, 0, 1450), (0, 60, 1580), (0, 0, 900), (180, 0, 1400), (460, 0, 1400), (720, 0, 1400), (130, 0, 900),
    (130, 0, 480), (130, 0, 60), (-180, 0, 1400), (-460, 0, 1400), (-720, 0, 1400), (-130, 0, 900),
    (-130, 0, 480), (-130, 0, 60), (35, 50, 1620), (75, 0, 1600), (-35, 50, 1620), (-75, 0, 1600)]
_OP = [  # 0 pelvis 1-3 R leg 4-6 L leg 7 thorax 8 head 9-11 L arm 12-14 R arm
    (0, 0, 900), (-130, 0, 900), (-130, 0, 480), (-130, 0, 60), (130, 0, 900), (130, 0, 480), (130, 0, 60),
    (0, 0, 1400), (0, 0, 1650), (180, 0, 1400), (460, 0, 1400), (720, 0, 1400), (-180, 0, 1400),
    (-460, 0, 1400), (-720, 0, 1400)]

DATASETS = {
    # name: (template, limb-end joints scaled by scaling_modifier (gaussian_model.py:173-178),
    #        limb pairs of utils/loss_utils.py:226-250 as (l_arm, r_arm, l_leg, r_leg), W, H, rasterizer key)
    "h36m": dict(template=_H36M, limb_ends=[3, 6, 12, 13, 15, 16],
                 limbs=((12, 13), (15, 16), (5, 6), (2, 3)), W=1000, H=1000, fx=1145.0, ring=5000.0,
                 rendering="diff-gaussian-rasterization-h36m"),
    "panoptic": dict(template=_PANOPTIC, limb_ends=[8, 14, 4, 5, 10, 11],
                     limbs=((4, 5), (10, 11), (7, 8), (13, 14)), W=1920, H=1080, fx=1400.0, ring=3000.0,
                     rendering="diff-gaussian-rasterization-panoptic"),
    "occlusion-person": dict(template=_OP, limb_ends=[3, 6, 10, 11, 13, 14],
                             limbs=((10, 11), (13, 14), (5, 6), (2, 3)), W=1280, H=720, fx=1145.0, ring=5000.0,
                             rendering="diff-gaussian-rasterization-op"),
}


def skeleton_template(dataset):
    return np.asarray(DATASETS[dataset]["template"], dtype=np.float64)


def ring_cameras(V, W, H, radius, fx, rng, height=1500.0, target=(0.0, 0.0, 900.0), device="cpu"):
    """SURVEY §8d synthetic cameras: ring of V cameras looking at the pelvis, jittered azimuth and principal point."""
    cams = []
    for k in range(V):
        az = 2 * math.pi * k / V + rng.uniform(-0.1, 0.1)
        pos = (radius * math.cos(az) + target[0], radius * math.sin(az) + target[1], height)
        cx = W / 2 + rng.uniform(-15, 15)
        cy = H / 2 + rng.uniform(-15, 15)
        cams.append(look_at_camera(k, pos, target, fx, fx, cx, cy, W, H, device=device))
    return cams


def project_points(cam, pts):
    """Pixel coordinates (x, y) of world points through K[R|t] (as triangulation.py:59-67 builds it)."""
    Rw2c = cam.R.T
    pc = (Rw2c @ np.asarray(pts, dtype=np.float64).T).T + cam.T[None]
    uv = (cam.K @ pc.T).T
    return uv[:, :2] / uv[:, 2:3]


class SyntheticScene:
    """One synthetic frame: GT skeleton, noisy initial guess, cameras, noisy 2D keypoints (seed-reproducible)."""

    def __init__(self, dataset="h36m", n_views=4, seed=0, device="cpu", W=None, H=None, n_skeletons=1,
                 pitch=1500.0, ring=None, fx=None):
        d = DATASETS[dataset]
        rng = np.random.default_rng(seed)
        self.dataset = dataset
        self.W = int(W or d["W"])
        self.H = int(H or d["H"])
        tmpl = skeleton_template(dataset)
        J = tmpl.shape[0]
        gts = []
        side = int(math.ceil(math.sqrt(n_skeletons)))
        for s in range(n_skeletons):
            off = np.array([((s % side) - (side - 1) / 2) * pitch, ((s // side) - (side - 1) / 2) * pitch, 0.0])
            gts.append(tmpl + rng.normal(0.0, 50.0, tmpl.shape) + off)
        self.pose_3d_gt = np.concatenate(gts, 0)
        self.pose_3d_init = self.pose_3d_gt + rng.normal(0.0, 30.0, self.pose_3d_gt.shape)
        self.n_joints = J
        self.n_points = J * n_skeletons
        fx = fx or d["fx"] * (self.W / d["W"])
        self.cameras = ring_cameras(n_views, self.W, self.H, ring or d["ring"], fx, rng, device=device)
        self.poses_2d = np.stack([project_points(c, self.pose_3d_gt) + rng.normal(0.0, 3.0, (self.n_points, 2))
                                  for c in self.cameras], 0)
        self.spatial_lr_scale = cameras_extent(self.cameras)


def random_gaussian_params(scene, seed, scale_log=4.0, rand_rot=True, opac=None, onehot=False):
    """Seeded rasterizer inputs for `scene`'s points that exercise every term (tests, the stress benchmark): anisotropic
    scales exp(N(scale_log, 0.3)), random unit quaternions (or identity), opacities U(0.3, 1) (or a constant), features
    one-hot by joint index (+ U(0, 0.1) on every channel unless `onehot`).  numpy fp32 arrays: means (P,3), scales (P,3),
    quats (P,4), opac (P,1), feat (P,C)."""
    rng = np.random.default_rng(seed + 100)
    P, C = scene.n_points, scene.n_joints
    out = {"means": scene.pose_3d_init.astype(np.float32)}
    out["scales"] = np.exp(rng.normal(scale_log, 0.3, (P, 3))).astype(np.float32)
    q = rng.normal(0, 1, (P, 4)) if rand_rot else np.tile([1.0, 0, 0, 0], (P, 1))
    out["quats"] = (q / np.linalg.norm(q, axis=1, keepdims=True)).astype(np.float32)
    out["opac"] = (np.full((P, 1), opac, np.float32) if opac is not None else rng.uniform(0.3, 1.0, (P, 1)).astype(np.float32))
    eye = np.eye(C, dtype=np.float32)[np.arange(P) % C]
    out["feat"] = eye if onehot else (eye + 0.1 * rng.uniform(0, 1, (P, C))).astype(np.float32)
    out["_rng"] = rng      # (callers that draw more -- upstream gradients -- continue the same stream)
    return out


def stress_scene(n_views=8, seed=42, W=2048, H=2048):
    """BASELINE configs[4]: 256 skeletons on a 16 x 16 grid of 1 500 mm pitch (P = 4 352, C = 17), ring cameras at 20 m,
    fx = fy = 2 300, one-hot features, opacity 1.  Returns (scene, params dict of random_gaussian_params)."""
    sc = SyntheticScene("h36m", n_views=n_views, seed=seed, W=W, H=H, ring=20000.0, fx=2300.0, n_skeletons=256, pitch=1500.0)
    return sc, random_gaussian_params(sc, seed, scale_log=3.0, onehot=True, opac=1.0)


# ------------------------------------------------------------------------------------------------------------
# Gaussian parameter container (reference: scene/gaussian_model.py:32-47, 102-131, 149-200, 203-248)
# ------------------------------------------------------------------------------------------------------------


def inverse_sigmoid(x):
    return torch.log(x / (1 - x))


class ExponentialLR:
    """The xyz learning-rate schedule (the reference's get_expon_lr_func, utils/general_utils.py:38-71), written in the
    closed form the device-side optimiser evaluates (csrc/sks_loop_dev.h, adam_block_begin): geometric interpolation
    lr_init^(1-t) * lr_final^t over t = step / max_steps clamped to [0, 1], times a sine warm-up from `lr_delay_mult` to 1
    over the first `lr_delay_steps`; zero for a negative step or an all-zero schedule.  Host and device hold the same two
    logarithms, so `sks_loop_adam_step` and torch.optim.Adam driven by this object take the same step sizes (pinned to the
    reference's values by tests/golden/reference_python.npz)."""

    def __init__(self, lr_init, lr_final, lr_delay_steps=0, lr_delay_mult=1.0, max_steps=1000000):
        self.off = lr_init == 0.0 and lr_final == 0.0
        # np.log(0) = -inf in the reference (general_utils.py:66): a schedule with ONE zero end point is 0 wherever that end
        # point has weight (exp(-inf)) and NaN where its weight is exactly 0 (0 * -inf); IEEE arithmetic below gives the same
        self.log_init = math.log(lr_init) if lr_init > 0.0 else (-math.inf if lr_init == 0.0 else math.nan)
        self.log_final = math.log(lr_final) if lr_final > 0.0 else (-math.inf if lr_final == 0.0 else math.nan)
        self.delay_steps, self.delay_mult, self.max_steps = lr_delay_steps, lr_delay_mult, max_steps

    def __call__(self, step):
        if self.off or step < 0:
            return 0.0
        warm = 1.0
        if self.delay_steps > 0:
            c = min(max(step / self.delay_steps, 0.0), 1.0)
            warm = self.delay_mult + (1.0 - self.delay_mult) * math.sin(0.5 * math.pi * c)
        t = min(max(step / self.max_steps, 0.0), 1.0)
        return warm * math.exp(self.log_init * (1.0 - t) + self.log_final * t)


def get_expon_lr_func(lr_init, lr_final, lr_delay_steps=0, lr_delay_mult=1.0, max_steps=1000000):
    """Name and signature of utils/general_utils.py:38 (scene/gaussian_model.py:241-246 calls it); returns an ExponentialLR."""
    return ExponentialLR(lr_init, lr_final, lr_delay_steps, lr_delay_mult, max_steps)


class OptimizationParams:
    """Field names of configs/*.yaml `optimization:` (configs/h36m.yaml:50-75) so the YAMLs stay valid."""
    iterations = 500
    position_lr_init = 0.0005
    position_lr_final = 0.000005
    position_lr_delay_mult = 0.0
    position_lr_max_steps = 4000
    feature_lr = 0.0
    opacity_lr = 0.0
    scaling_lr = 0.005
    rotation_lr = 0.001


class GaussianModel:
    """Skeleton Gaussians: one per joint, one-hot J-channel feature (gaussian_model.py:149-200)."""

    def __init__(self, sh_degree=1, optimizer_type="default"):
        self.active_sh_degree = 0
        self.max_sh_degree = sh_degree
        self.optimizer_type = optimizer_type
        self.optimizer = None
        self.spatial_lr_scale = 1.0

    def create_from_points(self, points, spatial_lr_scale, n_joints, opacity_on=True, scaling=3.0,
                           scaling_modifier=1.0, scene_type="h36m", device="cpu"):
        self.spatial_lr_scale = spatial_lr_scale
        pts = torch.tensor(np.asarray(points)).float().to(device)
        P = pts.shape[0]
        ch = torch.arange(P, device=device) % n_joints
        features = torch.zeros(P, n_joints, device=device).scatter_(1, ch[:, None], 1.0)[:, None, :]  # (P,1,C)
        scales = torch.ones_like(pts) * scaling
        # gaussian_model.py:173-178 is an if / elif chain over the scene types "h36m", "panoptic", "occlusion-person"; any other
        # string falls through WITHOUT a modifier.  That includes the reference's own configs/h36m-occ.yaml: its data_root
        # "data/h36m-occ" makes scene_type "h36m-occ" (scene/__init__.py:40), so its scaling_modifier 1.25 is never applied
        # (held to the reference's run of that config: tests/golden/reference_loop.npz, case h36m_occ).
        ends = torch.tensor(DATASETS.get(scene_type, {}).get("limb_ends", []), device=device, dtype=torch.long)
        sel = (ch[:, None] == ends[None, :]).any(1)
        scales[sel] *= scaling_modifier
        rots = torch.zeros((P, 4), device=device)
        rots[:, 0] = 1
        opacities = inverse_sigmoid(1.0 * torch.ones((P, 1), dtype=torch.float, device=device))
        self._xyz = torch.nn.Parameter(pts.requires_grad_(True))
        self._features_dc = torch.nn.Parameter(features.contiguous().requires_grad_(False))
        self._features_rest = torch.nn.Parameter(torch.zeros(P, 0, n_joints, device=device).requires_grad_(False))
        self._scaling = torch.nn.Parameter(scales.requires_grad_(True))
        self._rotation = torch.nn.Parameter(rots.requires_grad_(True))
        self._opacity = torch.nn.Parameter(opacities.requires_grad_(opacity_on))
        self._initial = (scales.detach().clone(), rots.detach().clone(), opacities.detach().clone())
        return self

    def reset_from_points(self, points):
        """Next frame of the same sequence: the reference builds a fresh GaussianModel per scene (train.py:86-99);
        here the parameter tensors are re-initialised IN PLACE (same storage), so captured hipGraphs and every
        pointer held by a MultiViewLoop stay valid."""
        with torch.no_grad():
            pts = points if torch.is_tensor(points) else torch.as_tensor(np.asarray(points))
            self._xyz.copy_(pts.to(device=self._xyz.device, dtype=self._xyz.dtype))
            self._scaling.copy_(self._initial[0])
            self._rotation.copy_(self._initial[1])
            self._opacity.copy_(self._initial[2])
        return self

    def get_covariance(self, scaling_modifier=1.0):
        """scene/gaussian_model.py:33-37,131-134 (build_covariance_from_scaling_rotation + strip_symmetric,
        utils/general_utils.py:61-119): the 6 upper-triangle entries (xx, xy, xz, yy, yz, zz) of R S S^T R^T, what the
        renderer passes as cov3D_precomp when pipe.compute_cov3D_python is set (gaussian_renderer/__init__.py:80-86)."""
        from .heatmaps import covariance_from_scaling_rotation
        cov = covariance_from_scaling_rotation(self.get_scaling, self._rotation, scaling_modifier)
        return torch.stack([cov[:, 0, 0], cov[:, 0, 1], cov[:, 0, 2], cov[:, 1, 1], cov[:, 1, 2], cov[:, 2, 2]], dim=1)

    # setup_functions, scene/gaussian_model.py:26-47: the activations as attributes, the getters through them (:100-110)
    scaling_activation = staticmethod(torch.exp)
    opacity_activation = staticmethod(torch.sigmoid)
    rotation_activation = staticmethod(torch.nn.functional.normalize)
    get_xyz = property(lambda self: self._xyz)
    get_scaling = property(lambda self: self.scaling_activation(self._scaling))
    get_rotation = property(lambda self: self.rotation_activation(self._rotation))
    get_opacity = property(lambda self: self.opacity_activation(self._opacity))
    get_features = property(lambda self: self._features_dc)
    get_features_dc = property(lambda self: self._features_dc)
    get_features_rest = property(lambda self: self._features_rest)

    def training_setup(self, opt=OptimizationParams):
        groups = [
            {"params": [self._xyz], "lr": opt.position_lr_init * self.spatial_lr_scale, "name": "xyz"},
            {"params": [self._features_dc], "lr": opt.feature_lr, "name": "f_dc"},
            {"params": [self._features_rest], "lr": opt.feature_lr / 20.0, "name": "f_rest"},
            {"params": [self._opacity], "lr": opt.opacity_lr, "name": "opacity"},
            {"params": [self._scaling], "lr": opt.scaling_lr, "name": "scaling"},
            {"params": [self._rotation], "lr": opt.rotation_lr, "name": "rotation"},
        ]
        self.optimizer = torch.optim.Adam(groups, lr=0.0, eps=1e-15)
        # the same numbers for the device-side optimiser step of skelsplat_amd.loop (sks_loop_adam_step)
        self.opt_cfg = dict(lr_init=opt.position_lr_init * self.spatial_lr_scale,
                            lr_final=opt.position_lr_final * self.spatial_lr_scale,
                            lr_delay_mult=opt.position_lr_delay_mult, lr_delay_steps=0,
                            lr_max_steps=opt.position_lr_max_steps, lr_scaling=opt.scaling_lr,
                            lr_rotation=opt.rotation_lr, lr_opacity=opt.opacity_lr, betas=(0.9, 0.999), eps=1e-15)
        self.xyz_scheduler_args = get_expon_lr_func(
            lr_init=opt.position_lr_init * self.spatial_lr_scale, lr_final=opt.position_lr_final * self.spatial_lr_scale,
            lr_delay_mult=opt.position_lr_delay_mult, max_steps=opt.position_lr_max_steps)

    def update_learning_rate(self, iteration):
        for g in self.optimizer.param_groups:
            if g["name"] == "xyz":
                lr = self.xyz_scheduler_args(iteration)
                g["lr"] = lr
                return lr
