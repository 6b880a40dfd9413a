"""Pure-PyTorch differentiable restatement of the reference rasterizer forward (TEST INFRASTRUCTURE ONLY).

Second, independent oracle: the forward follows DGR/cuda_rasterizer/forward.cu:74-401 as dense tensor ops
(every Gaussian evaluated on every pixel, explicit (depth, index) ordering, same thresholds); the backward is
torch autograd.  It cross-checks the hand-derived backward formulas the C oracle restates from
DGR/cuda_rasterizer/backward.cu, and it is the "pure-PyTorch CPU rasterization" bench.py times as cpu_baseline.

Where the reference's analytic backward deliberately differs from the true derivative of its forward, the same
behaviour is reproduced with straight-through terms so autograd yields the reference's gradient:
  * alpha = min(0.99, o*G) back-propagates as if unclamped            (backward.cu:569,619)
  * the background term -T_final/(1-alpha) * <bg, dL_dpix>             (backward.cu:612-615)
  * frustum-clamped t.x / t.y get zero gradient, t.z sees them as constants (backward.cu:182-183,310-312)
"""
import math

import torch

BLOCK = 16


def _rect(px, py, radius, gx, gy):
    # auxiliary.h:45-55 -- float arithmetic then truncation toward zero
    r = radius.to(px.dtype)
    xmin = torch.clamp(torch.trunc((px - r) / BLOCK), 0, gx).to(torch.int64)
    ymin = torch.clamp(torch.trunc((py - r) / BLOCK), 0, gy).to(torch.int64)
    xmax = torch.clamp(torch.trunc((px + r + (BLOCK - 1)) / BLOCK), 0, gx).to(torch.int64)
    ymax = torch.clamp(torch.trunc((py + r + (BLOCK - 1)) / BLOCK), 0, gy).to(torch.int64)
    return xmin, ymin, xmax, ymax


def rasterize(means3D, means2D, features, opacities, scales, rotations, cov3D_precomp, viewmatrix, projmatrix,
              W, H, tanfovx, tanfovy, bg=None, scale_modifier=1.0, antialiasing=False, dtype=torch.float32,
              dense=False):
    """Returns (color (C,H,W), radii (P,), invdepth (1,H,W)); differentiable wrt means3D, means2D (NDC-scaled
    screen-space dummy, like `viewspace_points`), features, opacities, scales, rotations, cov3D_precomp."""
    dev = means3D.device
    f = lambda t: None if t is None else t.to(dtype)
    means3D, features, opacities = f(means3D), f(features), f(opacities).reshape(-1)
    scales, rotations, cov3D_precomp = f(scales), f(rotations), f(cov3D_precomp)
    V = f(viewmatrix).reshape(4, 4)   # transposed (column-major in memory): p_view = p_row @ V
    Pm = f(projmatrix).reshape(4, 4)
    P, C = features.shape
    gx, gy = (W + BLOCK - 1) // BLOCK, (H + BLOCK - 1) // BLOCK
    focal_x = W / (2.0 * tanfovx)
    focal_y = H / (2.0 * tanfovy)

    ones = torch.ones(P, 1, dtype=dtype, device=dev)
    ph = torch.cat([means3D, ones], 1)
    p_view = ph @ V[:, :3]
    p_hom = ph @ Pm
    p_w = 1.0 / (p_hom[:, 3] + 0.0000001)
    p_proj = p_hom[:, :3] * p_w[:, None]
    in_front = p_view[:, 2] > 0.2

    # computeCov3D (forward.cu:114-150), quaternion not normalised
    if cov3D_precomp is None:
        s = scale_modifier * scales
        r, x, y, z = rotations.unbind(1)
        R = torch.stack([
            1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
            2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
            2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], 1).reshape(P, 3, 3)
        # glm column-major ctor: the 9 numbers above are columns; as a math matrix R_math = that.T
        Rm = R.transpose(1, 2)
        M = torch.diag_embed(s) @ Rm        # glm "S * R" with S,R as math matrices
        Sigma = M.transpose(1, 2) @ M
    else:
        c6 = cov3D_precomp.reshape(P, 6)
        Sigma = torch.stack([c6[:, 0], c6[:, 1], c6[:, 2], c6[:, 1], c6[:, 3], c6[:, 4], c6[:, 2], c6[:, 4], c6[:, 5]],
                            1).reshape(P, 3, 3)

    # computeCov2D (forward.cu:74-109)
    t = p_view
    tz = t[:, 2]
    limx, limy = 1.3 * tanfovx, 1.3 * tanfovy
    txtz, tytz = t[:, 0] / tz, t[:, 1] / tz
    xin = ((txtz >= -limx) & (txtz <= limx)).to(dtype)
    yin = ((tytz >= -limy) & (tytz <= limy)).to(dtype)
    txc_val = torch.clamp(txtz, -limx, limx) * tz
    tyc_val = torch.clamp(tytz, -limy, limy) * tz
    txc = t[:, 0] * xin + (txc_val - t[:, 0] * xin).detach()
    tyc = t[:, 1] * yin + (tyc_val - t[:, 1] * yin).detach()
    zero = torch.zeros_like(tz)
    # J as a math matrix (rows): [[fx/tz, 0, -fx*tx/tz^2], [0, fy/tz, -fy*ty/tz^2], [0,0,0]]
    J = torch.stack([focal_x / tz, zero, -(focal_x * txc) / (tz * tz),
                     zero, focal_y / tz, -(focal_y * tyc) / (tz * tz),
                     zero, zero, zero], 1).reshape(P, 3, 3)
    Wm = V[:3, :3].T            # world->view rotation as a math matrix
    JW = J @ Wm                 # (P,3,3)
    cov2 = JW @ Sigma @ JW.transpose(1, 2)
    cov_x, cov_y, cov_z = cov2[:, 0, 0], cov2[:, 0, 1], cov2[:, 1, 1]

    h_var = 0.3
    det_cov = cov_x * cov_z - cov_y * cov_y
    cov_x = cov_x + h_var
    cov_z = cov_z + h_var
    det = cov_x * cov_z - cov_y * cov_y
    h_scale = torch.ones_like(det)
    if antialiasing:
        h_scale = torch.sqrt(torch.clamp_min(det_cov / det, 0.000025))
    det_ok = det != 0
    det_inv = 1.0 / torch.where(det_ok, det, torch.ones_like(det))
    conic_x, conic_y, conic_z = cov_z * det_inv, -cov_y * det_inv, cov_x * det_inv

    with torch.no_grad():
        mid = 0.5 * (cov_x + cov_z)
        lam1 = mid + torch.sqrt(torch.clamp_min(mid * mid - det, 0.1))
        lam2 = mid - torch.sqrt(torch.clamp_min(mid * mid - det, 0.1))
        radius = torch.ceil(3.0 * torch.sqrt(torch.maximum(lam1, lam2)))
    pix_x = ((p_proj[:, 0] + 1.0) * W - 1.0) * 0.5
    pix_y = ((p_proj[:, 1] + 1.0) * H - 1.0) * 0.5
    with torch.no_grad():
        xmin, ymin, xmax, ymax = _rect(pix_x, pix_y, radius, gx, gy)
        visible = in_front & det_ok & ((xmax - xmin) * (ymax - ymin) > 0)
        radii = torch.where(visible, radius, torch.zeros_like(radius)).to(torch.int32)
    if means2D is not None:  # viewspace_points dummy: gradient is wrt NDC (backward.cu:527-528,626-627)
        pix_x = pix_x + means2D[:, 0].to(dtype) * (0.5 * W)
        pix_y = pix_y + means2D[:, 1].to(dtype) * (0.5 * H)
    opac = opacities * h_scale
    depth = p_view[:, 2]

    # ordering: stable sort of (tile, depth-bits) with index-major emission == per tile (depth, index)
    with torch.no_grad():
        dkey = depth.detach().to(torch.float32).contiguous().view(torch.int32).to(torch.int64)
        order = sorted(range(P), key=lambda i: (int(dkey[i]), i))

    ys = torch.arange(H, device=dev)
    xs = torch.arange(W, device=dev)
    tyi = (ys // BLOCK)[:, None]
    txi = (xs // BLOCK)[None, :]
    pyf = ys.to(dtype)[:, None]
    pxf = xs.to(dtype)[None, :]

    T = torch.ones(H, W, dtype=dtype, device=dev)
    done = torch.zeros(H, W, dtype=torch.bool, device=dev)
    color = torch.zeros(C, H, W, dtype=dtype, device=dev)
    invd = torch.zeros(H, W, dtype=dtype, device=dev)
    bgC = torch.zeros(C, dtype=dtype, device=dev)
    if bg is not None:
        b = bg.to(dtype).reshape(-1)
        bgC[:min(C, b.numel())] = b[:C]
    vis_l = visible.tolist()
    xmin_l, ymin_l, xmax_l, ymax_l = xmin.tolist(), ymin.tolist(), xmax.tolist(), ymax.tolist()
    for g in order:
        if not vis_l[g]:
            continue
        if dense:   # every pixel evaluated, tile rect as a mask (slow; cross-check of the sliced form)
            sy, sx = slice(0, H), slice(0, W)
            in_rect = (tyi >= ymin[g]) & (tyi < ymax[g]) & (txi >= xmin[g]) & (txi < xmax[g])
        else:       # only the pixels of the Gaussian's tile rect (what a CPU rasterizer would do)
            sy = slice(ymin_l[g] * BLOCK, min(H, ymax_l[g] * BLOCK))
            sx = slice(xmin_l[g] * BLOCK, min(W, xmax_l[g] * BLOCK))
            in_rect = True
        Ts, dones = T[sy, sx], done[sy, sx]
        dx = pix_x[g] - pxf[:, sx]
        dy = pix_y[g] - pyf[sy]
        power = -0.5 * (conic_x[g] * dx * dx + conic_z[g] * dy * dy) - conic_y[g] * dx * dy
        G = torch.exp(torch.clamp_max(power, 0.0))
        a_raw = opac[g] * G
        alpha = a_raw + (torch.clamp_max(a_raw, 0.99) - a_raw).detach()
        valid = (power.detach() <= 0) & (alpha.detach() >= 1.0 / 255.0) & ~dones
        if dense:
            valid = valid & in_rect
        test_T = Ts * (1 - alpha)
        newly_done = valid & (test_T.detach() < 0.0001)
        contrib = valid & ~newly_done
        w = torch.where(contrib, alpha * Ts, torch.zeros_like(Ts))
        pad = (sx.start, W - sx.stop, sy.start, H - sy.stop)
        color = color + torch.nn.functional.pad(features[g][:, None, None] * w[None], pad)
        invd = invd + torch.nn.functional.pad((1.0 / depth[g]) * w, pad)
        T = T + torch.nn.functional.pad(torch.where(contrib, test_T - Ts, torch.zeros_like(Ts)), pad)
        done = done | torch.nn.functional.pad(newly_done, pad)
    # straight-through background term (value unchanged: the reference does not composite bg, forward.cu:396)
    bgterm = T[None] * bgC[:, None, None]
    color = color + (bgterm - bgterm.detach())
    return color, radii, invd[None]


def l2_loss_gaussian(rendering, gt_heatmap):
    """utils/loss_utils.py:86-100 (value only, 'mean' reduction)."""
    mask = (gt_heatmap > 0) | (rendering > 0)
    err = (rendering - gt_heatmap) ** 2
    return err[mask].mean()
