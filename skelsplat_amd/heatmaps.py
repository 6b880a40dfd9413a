"""Pseudo-GT heat-maps (reference: utils/general_utils.py:175-304 generate_heatmaps + normalize_heatmaps), HIP only.

The reference writes a single 255 impulse at (floor(y), floor(x)) of each joint channel and runs
cupyx.scipy.ndimage.gaussian_filter(sigma=[sqrt(lambda1), sqrt(lambda2)]) over the full-resolution plane, V*J times
per scene, then min-max normalises each channel.  Filtering an impulse is closed form: the result is
255 * outer(k_rows, k_cols) with k the (truncate = 4 sigma, sum-normalised, 'reflect'-extended) 1-D kernels, so a
frame's heat-maps are two launches: sks_heatmap_factors (lambda1/lambda2 by the reference's own transcription of the
EWA projection -- (R J)^T Sigma^T (R J), not the rasterizer's footprint, a quirk a drop-in keeps -- the 1-D responses and
the min-max constants) and sks_heatmaps (all planes in one streaming write).  There is no CPU path here: the
tensor-op restatement that pins this against the reference's own output lives in oracle/heatmaps_ref.py (tests only).
"""
import torch


def covariance_from_scaling_rotation(scaling, rotation_raw, modifier=1.0):
    """scene/gaussian_model.py:33-37 + general_utils.py:87-119: Sigma = R S S^T R^T with the *normalised* quaternion."""
    q = rotation_raw / rotation_raw.norm(dim=1, keepdim=True)
    r, x, y, z = q.unbind(1)
    R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
                     2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
                     2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], 1).reshape(-1, 3, 3)
    L = R @ torch.diag_embed(modifier * scaling)
    return L @ L.transpose(1, 2)


def heatmap_factors(means, scaling, rotation_raw, poses_2d, cameras, scaling_modifier=1.0, views=None, frames=1, out=None,
                    drop_mask=None):
    """The separable description of the (V, J, H, W) heat-maps: row (V,J,H) = 255 * impulse response along y,
    col (V,J,W) = impulse response along x, cmin (V,J), den (V,J) with
    plane = (row[:, None] * col[None, :] - cmin) / den.
    The plane minimum / maximum are the products of the factor minima / maxima (everything is non-negative and fp32
    multiplication is monotone), so the min-max normalisation of normalize_heatmaps (:300-304) needs no image pass.
    One kernel launch (sks_heatmap_factors), no host synchronisation.  `views`: a rasterizer.ViewBatch of `cameras` to
    reuse (scene streaming), else built here.  `frames` > 1: independent frames in one launch -- `cameras` / `views`
    list frames x Vf views frame-major, the parameters are stacked (frames, J, ..) and poses_2d is (frames*Vf, J, 2).
    `out`: a rasterizer.HeatmapFactors to fill in place (the sparse fused step reads the factors themselves, no plane is
    ever written); only then may the cameras differ in size (rows / columns keep the largest view's strides).
    `drop_mask` (V,J) bool: planes without an impulse (see generate_heatmaps)."""
    from . import _lib
    from .rasterizer import ViewBatch, _f32c
    dev = means.device
    V = len(cameras)
    frames = int(frames)
    if frames < 1 or V % frames or (frames > 1 and (means.dim() != 3 or means.shape[0] != frames)):
        raise ValueError(f"frames = {frames}: needs frames x Vf cameras and parameters stacked (frames, J, ..)")
    if views is None:
        views = ViewBatch.from_cameras(cameras, allow_mixed=out is not None)
    W, H = views.W, views.H
    if views.mixed and out is None:
        raise ValueError("generate_heatmaps: all cameras must share (W, H)")
    means, scaling, rotation_raw = _f32c(means, "means"), _f32c(scaling, "scaling"), _f32c(rotation_raw, "rotation")
    poses_2d = torch.as_tensor(poses_2d, device=dev)
    J = means.shape[-2]
    p2d = poses_2d.to(torch.float32).contiguous()
    if tuple(p2d.shape) != (V, J, 2):
        raise ValueError(f"poses_2d must be (V, J, 2) = {(V, J, 2)}, got {tuple(p2d.shape)}")
    if out is not None:
        if (out.V, out.J, out.W, out.H) != (V, J, W, H) or out.row.device != dev:
            raise ValueError(f"heatmap_factors: `out` is {(out.V, out.J, out.W, out.H)}, the views need {(V, J, W, H)}")
        row, col, cmin, den = out.row, out.col, out.cmin, out.den
    else:
        row = torch.empty((V, J, H), dtype=torch.float32, device=dev)
        col = torch.empty((V, J, W), dtype=torch.float32, device=dev)
        cmin = torch.empty((V, J), dtype=torch.float32, device=dev)
        den = torch.empty((V, J), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        rc = _lib.load().sks_heatmap_factors(V, J, W, H, means.data_ptr(), scaling.data_ptr(), rotation_raw.data_ptr(),
                                             float(scaling_modifier), p2d.data_ptr(), views.viewmatrix.data_ptr(),
                                             views.tanfovx, views.tanfovy, row.data_ptr(), col.data_ptr(),
                                             cmin.data_ptr(), den.data_ptr(), frames, views.wh,
                                             torch.cuda.current_stream(dev).cuda_stream)
    _lib.check(rc, "sks_heatmap_factors")
    if drop_mask is not None:
        dm = torch.as_tensor(drop_mask, dtype=torch.bool).to(row.device)
        row.masked_fill_(dm[:, :, None], 0.0)
        cmin.masked_fill_(dm, 0.0)
        den.masked_fill_(dm, 1.0)
    return row, col, cmin, den


def draw_dropout(n_views, n_joints, generator=None):
    """The reference's heat-map dropout (general_utils.py:267-283, `training.dropout`): three camera indices from
    torch.randint(4, (3,)) and three joint indices from torch.randint(n_joints, (3,)) -- the same two draws from the same
    (default, CPU) generator -- and in those cameras those joints get NO impulse, so their planes stay zero.  Returns the
    (n_views, n_joints) bool mask of dropped planes."""
    cams = torch.randint(4, (3,), generator=generator)
    joints = torch.randint(n_joints, (3,), generator=generator)
    mask = torch.zeros((n_views, n_joints), dtype=torch.bool)
    for c in cams.tolist():
        if c < n_views:
            mask[c, joints] = True
    return mask


def generate_heatmaps(means, scaling, rotation_raw, poses_2d, cameras, scaling_modifier=1.0, out=None, views=None,
                      totals=None, dropout=False, drop_mask=None, frames=1):
    """(V, J, H, W) normalised heat-maps on the parameters' ROCm device; all cameras must share (W, H).  poses_2d:
    (V, J, 2) pixel (x, y).  general_utils.py:175-304 with dropout=False, as two launches (sks_heatmap_factors, then the
    planes by the streaming kernel sks_heatmaps).  `out`: optional (V,J,H,W) fp32 buffer to write into (scene
    streaming: same storage for every frame); `totals`: optional (V,2) fp64 tensor that receives each view's
    {sum gt^2, count gt > 0} (rasterizer.GtStats.totals) while the planes are written, instead of a separate pass.
    `dropout=True` draws the reference's dropped (camera, joint) planes (draw_dropout); `drop_mask` (V,J) bool gives them
    explicitly.  A dropped plane has no impulse: it is all zero before and after normalize_heatmaps (0 / 1e-8), which the
    separable form states as row = 0, cmin = 0, den = 1.  `frames`: see heatmap_factors."""
    from . import _lib
    if dropout and drop_mask is None:
        drop_mask = draw_dropout(len(cameras), means.shape[-2])
    row, col, cmin, den = heatmap_factors(means, scaling, rotation_raw, poses_2d, cameras, scaling_modifier, views=views,
                                          frames=frames, drop_mask=drop_mask)
    V, J, H = row.shape
    W = col.shape[2]
    if out is None:
        out = torch.empty((V, J, H, W), dtype=torch.float32, device=row.device)
    elif out.shape != (V, J, H, W) or out.dtype != torch.float32 or not out.is_contiguous() or out.device != row.device:
        raise ValueError("generate_heatmaps: `out` must be a contiguous fp32 (V,J,H,W) tensor on the parameters' device")
    if totals is not None and (totals.shape != (V, 2) or totals.dtype != torch.float64 or not totals.is_contiguous()
                               or totals.device != row.device):
        raise ValueError("generate_heatmaps: `totals` must be a contiguous fp64 (V,2) tensor on the parameters' device")
    with torch.cuda.device(row.device):
        rc = _lib.load().sks_heatmaps(V, J, W, H, row.data_ptr(), col.data_ptr(), cmin.data_ptr(), den.data_ptr(),
                                      out.data_ptr(), _lib.ptr(totals), torch.cuda.current_stream(row.device).cuda_stream)
    _lib.check(rc, "sks_heatmaps")
    return out
