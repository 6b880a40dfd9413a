"""Pseudo-GT heat-maps (reference: utils/general_utils.py:175-304 generate_heatmaps + normalize_heatmaps).

The reference writes a single 255 impulse at (floor(y), floor(x)) of each joint channel and runs
cupyx.scipy.ndimage.gaussian_filter(sigma=[sqrt(lambda1), sqrt(lambda2)]) over the full-resolution plane, V*J times
per scene, then min-max normalises each channel.  Filtering an impulse is closed form: the result is
255 * outer(k_rows, k_cols) with k the (truncate = 4 sigma, sum-normalised, 'reflect'-extended) 1-D kernels, so all
V*J planes are produced by a handful of small tensor ops.  lambda1/lambda2 come from the same EWA projection the
rasterizer uses (axis-aligned: the reference ignores the eigen-directions, :252-265, 287-289).
"""
import math

import torch


def ewa_lambdas(means, cov3D, cam, W, H):
    """(lambda1, lambda2) of every Gaussian in one camera: general_utils.py:201-265 (same math as
    DGR/cuda_rasterizer/forward.cu:74-109, 219-243, including the +0.3 px^2 low-pass and the max(0.1, .) guard)."""
    dt = torch.float32
    means = means.to(dt)
    P = means.shape[0]
    Vt = cam.world_view_transform.to(device=means.device, dtype=dt)     # transposed view matrix
    tanx, tany = math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5)
    fx, fy = W / (2.0 * tanx), H / (2.0 * tany)
    ph = torch.cat([means, torch.ones(P, 1, dtype=dt, device=means.device)], 1)
    t = ph @ Vt[:, :3]
    tz = t[:, 2]
    tx = torch.clamp(t[:, 0] / tz, -1.3 * tanx, 1.3 * tanx) * tz
    ty = torch.clamp(t[:, 1] / tz, -1.3 * tany, 1.3 * tany) * tz
    z = torch.zeros_like(tz)
    J = torch.stack([fx / tz, z, -(fx * tx) / (tz * tz), z, fy / tz, -(fy * ty) / (tz * tz), z, z, z], 1).reshape(P, 3, 3)
    Wm = Vt[:3, :3].T
    JW = J @ Wm
    cov = JW @ cov3D.to(dt) @ JW.transpose(1, 2)
    cx, cy, cz = cov[:, 0, 0] + 0.3, cov[:, 0, 1], cov[:, 1, 1] + 0.3
    det = cx * cz - cy * cy
    mid = 0.5 * (cx + cz)
    root = torch.sqrt(torch.clamp_min(mid * mid - det, 0.1))
    return mid + root, mid - root


def covariance_from_scaling_rotation(scaling, rotation_raw, modifier=1.0):
    """scene/gaussian_model.py:33-37 + general_utils.py:87-119: Sigma = R S S^T R^T with the *normalised* quaternion."""
    q = rotation_raw / rotation_raw.norm(dim=1, keepdim=True)
    r, x, y, z = q.unbind(1)
    R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
                     2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
                     2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], 1).reshape(-1, 3, 3)
    L = R @ torch.diag_embed(modifier * scaling)
    return L @ L.transpose(1, 2)


def _impulse_response_1d(pos, sigma, n, device):
    """scipy.ndimage.gaussian_filter1d of a unit impulse at integer `pos` on a length-n axis, mode='reflect',
    truncate=4.0.  pos, sigma: (K,) tensors.  Returns (K, n)."""
    K = pos.shape[0]
    idx = torch.arange(n, device=device, dtype=torch.float64)[None, :]
    sig = sigma.to(torch.float64)[:, None]
    p = pos.to(torch.float64)[:, None]
    radius = torch.floor(4.0 * sig + 0.5)
    # kernel normalisation: sum over j = -radius..radius of exp(-0.5 j^2 / sigma^2)
    rmax = int(radius.max().item()) if K else 0
    jj = torch.arange(-rmax, rmax + 1, device=device, dtype=torch.float64)[None, :]
    wj = torch.exp(-0.5 * jj * jj / (sig * sig)) * (jj.abs() <= radius)
    norm = wj.sum(1, keepdim=True)

    def tap(src):  # contribution of the image of the impulse at (possibly mirrored) coordinate `src`
        d = idx - src
        return torch.exp(-0.5 * d * d / (sig * sig)) * (d.abs() <= radius)

    # 'reflect' extension (d c b a | a b c d | d c b a): mirrors of p about -0.5 and n-0.5 (one bounce each side
    # is enough while radius < n, which holds for every realistic sigma)
    out = tap(p) + tap(-1.0 - p) + tap(2.0 * n - 1.0 - p)
    return (out / norm).to(torch.float32)


def generate_heatmaps(means, scaling, rotation_raw, poses_2d, cameras, scaling_modifier=1.0):
    """(V, J, H, W) normalised heat-maps; all cameras must share (W, H).  poses_2d: (V, J, 2) pixel (x, y).
    general_utils.py:175-304 with dropout=False."""
    dev = means.device
    V = len(cameras)
    W, H = int(cameras[0].image_width), int(cameras[0].image_height)
    cov3D = covariance_from_scaling_rotation(scaling, rotation_raw, scaling_modifier)
    poses_2d = torch.as_tensor(poses_2d, device=dev)
    out = []
    for v, cam in enumerate(cameras):
        l1, l2 = ewa_lambdas(means, cov3D, cam, W, H)
        xs = torch.clamp(poses_2d[v, :, 0].long(), 0, W - 1)   # .long() truncates like the reference (:275-278)
        ys = torch.clamp(poses_2d[v, :, 1].long(), 0, H - 1)
        ky = _impulse_response_1d(ys, torch.sqrt(l1), H, dev)  # sigma1 filters axis 0 (rows)
        kx = _impulse_response_1d(xs, torch.sqrt(l2), W, dev)  # sigma2 filters axis 1 (columns)
        hm = 255.0 * ky[:, :, None] * kx[:, None, :]
        cmin = hm.amin(dim=(1, 2), keepdim=True)
        cmax = hm.amax(dim=(1, 2), keepdim=True)
        out.append((hm - cmin) / (cmax - cmin + 1e-8))          # normalize_heatmaps (:300-304)
    return torch.stack(out, 0)
