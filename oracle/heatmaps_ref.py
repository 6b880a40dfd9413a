"""TEST INFRASTRUCTURE -- tensor-op restatement of the reference's pseudo-GT heat-map generator
(utils/general_utils.py:175-304 generate_heatmaps + normalize_heatmaps), runs on CPU (or any torch device).

Only tests import this.  It is pinned against the reference itself (tests/golden/reference_heatmaps.npz, produced by the
reference's own function in the build container) and against scipy.ndimage.gaussian_filter; the product
(skelsplat_amd/heatmaps.py -> sks_heatmap_factors + sks_heatmaps, HIP only) is then held to it on the GPU.

The reference writes a single 255 impulse at (floor(y), floor(x)) of each joint channel and runs
cupyx.scipy.ndimage.gaussian_filter(sigma=[sqrt(lambda1), sqrt(lambda2)]) over the full-resolution plane, V*J times
per scene, then min-max normalises each channel.  Filtering an impulse is closed form: the result is
255 * outer(k_rows, k_cols) with k the (truncate = 4 sigma, sum-normalised, 'reflect'-extended) 1-D kernels.
lambda1/lambda2 come from the reference's own transcription of the EWA projection (see ewa_lambdas_views: not the
rasterizer's footprint) and are used axis-aligned: the reference ignores the eigen-directions (:252-265, 287-289).
"""
import math

import torch


def ewa_lambdas(means, cov3D, cam, W, H):
    """(lambda1, lambda2) of every Gaussian in one camera: general_utils.py:189-265; see ewa_lambdas_views."""
    l1, l2 = ewa_lambdas_views(means, cov3D, [cam], W, H)
    return l1[0], l2[0]


def ewa_lambdas_views(means, cov3D, cameras, W, H):
    """(lambda1, lambda2), each (V, P): utils/general_utils.py:189-265, restated LITERALLY.

    The reference transcribes the rasterizer's glm expressions (forward.cu:74-109: T = W * J,
    cov = transpose(T) * transpose(Vrk) * T) into torch calls with the same operand order, but torch matrices are
    row-major where glm's constructors fill columns: with J the Jacobian (rows = d(screen)/d(camera)) and R the
    camera rotation, the rasterizer's 2D covariance is (J R) Sigma (J R)^T while this one is (R J)^T Sigma^T (R J).
    They differ whenever R is not the identity, so the pseudo-GT blobs are NOT the projected Gaussians' footprints --
    a quirk of the reference that a drop-in has to keep (pinned by tests/golden/reference_heatmaps.npz, produced by
    the reference's own function).  +0.3 px^2 low-pass and max(0.1, .) guard as in the rasterizer."""
    dt = torch.float32
    dev = means.device
    means = means.to(dt)
    P = means.shape[0]
    view_matrix = torch.stack([cam.world_view_transform.to(device=dev, dtype=dt).T for cam in cameras], 0)   # (V,4,4)
    tanx = torch.tensor([math.tan(cam.FoVx * 0.5) for cam in cameras], dtype=dt, device=dev)[:, None]
    tany = torch.tensor([math.tan(cam.FoVy * 0.5) for cam in cameras], dtype=dt, device=dev)[:, None]
    fx = W / (2.0 * tanx)
    fy = H / (2.0 * tany)
    hom = torch.cat([means, torch.ones(P, 1, dtype=dt, device=dev)], 1)
    t = torch.matmul(view_matrix, hom.T).transpose(1, 2)[:, :, :3]                     # (V,P,3) camera-space means
    tz = t[..., 2]
    tx = torch.minimum(torch.maximum(t[..., 0] / tz, -1.3 * tanx), 1.3 * tanx) * tz
    ty = torch.minimum(torch.maximum(t[..., 1] / tz, -1.3 * tany), 1.3 * tany) * tz
    z = torch.zeros_like(tz)
    J = torch.stack([fx / tz, z, -(fx * tx) / tz ** 2, z, fy / tz, -(fy * ty) / tz ** 2, z, z, z], -1).reshape(-1, P, 3, 3)
    Wm = view_matrix[:, :3, :3].unsqueeze(1)                                             # (V,1,3,3)
    T = Wm @ J
    cov = T.permute(0, 1, 3, 2) @ cov3D.to(dt).permute(0, 2, 1)[None] @ T
    cx, cy, cz = cov[..., 0, 0] + 0.3, cov[..., 0, 1], cov[..., 1, 1] + 0.3
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


def heatmap_factors(means, scaling, rotation_raw, poses_2d, cameras, scaling_modifier=1.0):
    """The separable description of the (V, J, H, W) heat-maps: row (V,J,H) = 255 * impulse response along y,
    col (V,J,W) = impulse response along x, cmin (V,J), den (V,J) with
    plane = (row[:, None] * col[None, :] - cmin) / den.
    The plane minimum / maximum are the products of the factor minima / maxima (everything is non-negative and fp32
    multiplication is monotone), so the min-max normalisation of normalize_heatmaps (:300-304) needs no image pass."""
    dev = means.device
    W, H = int(cameras[0].image_width), int(cameras[0].image_height)
    V = len(cameras)
    poses_2d = torch.as_tensor(poses_2d, device=dev)
    cov3D = covariance_from_scaling_rotation(scaling, rotation_raw, scaling_modifier)
    l1, l2 = ewa_lambdas_views(means, cov3D, cameras, W, H)           # (V, J) each
    J = l1.shape[1]
    xs = torch.clamp(poses_2d[:, :, 0].long(), 0, W - 1)              # .long() truncates like the reference (:275-278)
    ys = torch.clamp(poses_2d[:, :, 1].long(), 0, H - 1)
    # sigma1 filters axis 0 (rows), sigma2 axis 1 (columns); all V*J one-dimensional responses in one go
    row = (255.0 * _impulse_response_1d(ys.reshape(-1), torch.sqrt(l1).reshape(-1), H, dev)).reshape(V, J, H).contiguous()
    col = _impulse_response_1d(xs.reshape(-1), torch.sqrt(l2).reshape(-1), W, dev).reshape(V, J, W).contiguous()
    cmin = row.amin(dim=2) * col.amin(dim=2)
    cmax = row.amax(dim=2) * col.amax(dim=2)
    den = cmax - cmin + 1e-8
    return row, col, cmin.contiguous(), den.contiguous()



def generate_heatmaps(means, scaling, rotation_raw, poses_2d, cameras, scaling_modifier=1.0, totals=None):
    """(V, J, H, W) normalised heat-maps; general_utils.py:175-304 with dropout=False.  `totals`: optional (V,2) fp64
    tensor receiving each view's {sum gt^2, count gt > 0}."""
    row, col, cmin, den = heatmap_factors(means, scaling, rotation_raw, poses_2d, cameras, scaling_modifier)
    out = (row[:, :, :, None] * col[:, :, None, :] - cmin[:, :, None, None]) / den[:, :, None, None]
    if totals is not None:
        totals[:, 0] = (out.double() ** 2).sum(dim=(1, 2, 3))
        totals[:, 1] = (out > 0).double().sum(dim=(1, 2, 3))
    return out
