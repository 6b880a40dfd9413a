"""Python surface of the smaller hot-path ops (C ABI: sks_masked_l2, sks_fused_ssim_*, sks_knn3_meandist2*)."""
import torch

from . import _lib
from .rasterizer import _need_gpu


def _chk(t, name):
    _need_gpu(t, name)
    if t.dtype != torch.float32:
        raise RuntimeError(f"skelsplat_amd: `{name}` must be float32 (got {t.dtype})")
    return t.contiguous()


def masked_l2(render, gt, want_grad=True):
    """Fused `l2_loss_gaussian` (utils/loss_utils.py:86-100) for V views: render, gt (V,C,H,W).
    Returns (dL_unscaled or None, S (V,) f64, N (V,) f64): loss_v = S_v / N_v and dloss_v/drender = dL_unscaled / N_v."""
    render, gt = _chk(render, "render"), _chk(gt, "gt")
    if render.shape != gt.shape:
        raise RuntimeError(f"render {tuple(render.shape)} and gt {tuple(gt.shape)} differ")
    V = render.shape[0]
    n = render[0].numel()
    dev = render.device
    dL = torch.empty_like(render) if want_grad else None
    sums = torch.empty((V, 2), dtype=torch.float64, device=dev)
    with torch.cuda.device(dev):
        rc = _lib.load().sks_masked_l2(V, n, render.data_ptr(), gt.data_ptr(), _lib.ptr(dL), sums.data_ptr(),
                                       torch.cuda.current_stream(dev).cuda_stream)
    _lib.check(rc, "sks_masked_l2")
    return dL, sums[:, 0], sums[:, 1]


def masked_l2_grad_fused(render, gt):
    """Drop-in for loop.masked_l2_grad_torch: (true gradient, per-view loss)."""
    dL, S, N = masked_l2(render, gt)
    n = N.clamp_min(1.0)
    return dL, (S / n).to(torch.float32), (1.0 / n).to(torch.float32)


class _FusedL2LossGaussian(torch.autograd.Function):
    """loss = mean (or sum) of (rendering - gt)^2 over {gt > 0 or rendering > 0}; one fused pass forward (sks_masked_l2:
    read both images, write 2 (r - g) on the mask, S and N in fp64), one scaled copy backward; no host sync."""

    @staticmethod
    def forward(ctx, rendering, gt, mean):
        r, g = _chk(rendering, "rendering"), _chk(gt, "gt_heatmap")
        if r.shape != g.shape:
            raise RuntimeError(f"rendering {tuple(r.shape)} and gt_heatmap {tuple(g.shape)} differ")
        dev = r.device
        dL = torch.empty_like(r)
        sums = torch.empty((1, 2), dtype=torch.float64, device=dev)
        out = torch.empty(2, dtype=torch.float32, device=dev)     # [loss, gradient scale]: N == 0 -> nan, like the mean of nothing
        with torch.cuda.device(dev):
            rc = _lib.load().sks_masked_l2_loss(1, r.numel(), r.data_ptr(), g.data_ptr(), dL.data_ptr(), sums.data_ptr(),
                                                out.data_ptr(), out.data_ptr() + 4, 1 if mean else 0,
                                                torch._C._cuda_getCurrentRawStream(dev.index))
        _lib.check(rc, "sks_masked_l2_loss")
        ctx.save_for_backward(dL, out)
        ctx.shape = rendering.shape
        return out[0]

    @staticmethod
    def backward(ctx, gout):
        dL, out = ctx.saved_tensors
        return (dL * (gout * out[1])).reshape(ctx.shape), None, None


def l2_loss_gaussian(rendering, gt_heatmap, gt_2d=None, lambda_loss=1.0, reduction="mean"):
    """Fused drop-in for the reference's criterion `l2_loss_gaussian` (utils/loss_utils.py:86-100; selected through
    `losses[training.loss_function]`, train.py:61,150): same arguments, same `(loss, error)` return for 'mean', where
    the dense `error` image -- which train.py never reads -- is None."""
    if reduction == "mean":
        return _FusedL2LossGaussian.apply(rendering, gt_heatmap, True), None
    if reduction == "sum":
        return _FusedL2LossGaussian.apply(rendering, gt_heatmap, False)
    mask = (gt_heatmap > 0) | (rendering > 0)            # 'none': the gathered vector, as tensor ops on the device
    return ((rendering - gt_heatmap) ** 2)[mask]


class FusedSSIMMap(torch.autograd.Function):
    """submodules/fused-ssim/fused_ssim/__init__.py:8-32."""

    @staticmethod
    def forward(ctx, C1, C2, img1, img2, padding="same", train=True):
        img1c, img2c = _chk(img1, "img1"), _chk(img2, "img2")
        if img1c.dim() != 4 or img1c.shape != img2c.shape:
            raise RuntimeError("fused_ssim expects two (B,CH,H,W) tensors of equal shape")
        B, CH, H, W = img1c.shape
        dev = img1c.device
        ssim_map = torch.empty_like(img1c)
        parts = [torch.empty_like(img1c) for _ in range(3)] if train else [None, None, None]
        with torch.cuda.device(dev):
            rc = _lib.load().sks_fused_ssim_fwd(B, CH, H, W, float(C1), float(C2), img1c.data_ptr(), img2c.data_ptr(),
                                                ssim_map.data_ptr(), _lib.ptr(parts[0]), _lib.ptr(parts[1]),
                                                _lib.ptr(parts[2]), torch.cuda.current_stream(dev).cuda_stream)
        _lib.check(rc, "sks_fused_ssim_fwd")
        if padding == "valid":
            ssim_map = ssim_map[:, :, 5:-5, 5:-5]
        emp = torch.empty(0, device=dev)
        ctx.save_for_backward(img1c.detach(), img2c, *(p if p is not None else emp for p in parts))
        ctx.C1, ctx.C2, ctx.padding, ctx.train = C1, C2, padding, train
        return ssim_map

    @staticmethod
    def backward(ctx, opt_grad):
        img1, img2, dm_dmu1, dm_dsigma1_sq, dm_dsigma12 = ctx.saved_tensors
        if not ctx.train:
            raise RuntimeError("fused_ssim was called with train=False: no backward state was kept")
        dL_dmap = opt_grad
        if ctx.padding == "valid":
            dL_dmap = torch.zeros_like(img1)
            dL_dmap[:, :, 5:-5, 5:-5] = opt_grad
        dL_dmap = dL_dmap.contiguous()
        B, CH, H, W = img1.shape
        dev = img1.device
        grad = torch.empty_like(img1)
        with torch.cuda.device(dev):
            rc = _lib.load().sks_fused_ssim_bwd(B, CH, H, W, float(ctx.C1), float(ctx.C2), img1.data_ptr(), img2.data_ptr(),
                                                dL_dmap.data_ptr(), dm_dmu1.data_ptr(), dm_dsigma1_sq.data_ptr(),
                                                dm_dsigma12.data_ptr(), grad.data_ptr(),
                                                torch.cuda.current_stream(dev).cuda_stream)
        _lib.check(rc, "sks_fused_ssim_bwd")
        return None, None, grad, None, None, None


_SSIM_SCRATCH = {}


def _ssim_scratch(dev, stream):
    """The self-clearing reduction scratch of sks_fused_ssim_mean, one per (device, stream), zeroed once."""
    key = (dev.index, stream)
    buf = _SSIM_SCRATCH.get(key)
    if buf is None:
        buf = torch.zeros(_lib.SKS_SSIM_SCRATCH_BYTES // 8, dtype=torch.float64, device=dev)
        # allocated during a hipGraph capture it lives in that graph's private pool (and its zero fill is part of the graph):
        # never shared with other graphs or eager code
        if not torch.cuda.is_current_stream_capturing():
            _SSIM_SCRATCH[key] = buf
    return buf


def reset_ssim_scratch():
    """Forget the cached reduction scratch: a call that failed half-way may have left partial sums in it."""
    _SSIM_SCRATCH.clear()


class FusedSSIMMean(torch.autograd.Function):
    """`FusedSSIMMap.apply(...).mean()` in one pass each way: the forward accumulates the (cropped) mean while it
    writes the three partial-derivative maps and never writes the SSIM map; the backward takes the scalar upstream
    gradient, so no (B,CH,H,W) gradient image is materialised, padded or read."""

    @staticmethod
    def forward(ctx, C1, C2, img1, img2, padding="same", train=True):
        img1c, img2c = _chk(img1, "img1"), _chk(img2, "img2")
        if img1c.dim() != 4 or img1c.shape != img2c.shape:
            raise RuntimeError("fused_ssim expects two (B,CH,H,W) tensors of equal shape")
        B, CH, H, W = img1c.shape
        dev = img1c.device
        crop = 5 if padding == "valid" else 0
        count = B * CH * max(H - 2 * crop, 0) * max(W - 2 * crop, 0)
        parts = [torch.empty_like(img1c) for _ in range(3)] if train else [None, None, None]
        mean = torch.empty((), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            stream = torch.cuda.current_stream(dev).cuda_stream
            rc = _lib.load().sks_fused_ssim_mean(B, CH, H, W, float(C1), float(C2), img1c.data_ptr(), img2c.data_ptr(), crop,
                                                 _lib.ptr(parts[0]), _lib.ptr(parts[1]), _lib.ptr(parts[2]),
                                                 _ssim_scratch(dev, stream).data_ptr(), mean.data_ptr(), stream)
        if rc != 0:
            reset_ssim_scratch()
        _lib.check(rc, "sks_fused_ssim_mean")
        emp = torch.empty(0, device=dev)
        ctx.save_for_backward(img1c.detach(), img2c, *(p if p is not None else emp for p in parts))
        ctx.crop, ctx.count, ctx.train = crop, count, train
        return mean

    @staticmethod
    def backward(ctx, g):
        img1, img2, dm_dmu1, dm_dsigma1_sq, dm_dsigma12 = ctx.saved_tensors
        if not ctx.train:
            raise RuntimeError("fused_ssim was called with train=False: no backward state was kept")
        B, CH, H, W = img1.shape
        dev = img1.device
        grad = torch.empty_like(img1)
        g = g.to(torch.float32).contiguous()   # d loss / d mean, one device scalar; d mean / d map = 1 / count
        with torch.cuda.device(dev):
            rc = _lib.load().sks_fused_ssim_bwd_uniform(B, CH, H, W, img1.data_ptr(), img2.data_ptr(), g.data_ptr(),
                                                        1.0 / max(ctx.count, 1), ctx.crop, dm_dmu1.data_ptr(),
                                                        dm_dsigma1_sq.data_ptr(), dm_dsigma12.data_ptr(), grad.data_ptr(),
                                                        torch.cuda.current_stream(dev).cuda_stream)
        _lib.check(rc, "sks_fused_ssim_bwd_uniform")
        return None, None, grad, None, None, None


allowed_padding = ["same", "valid"]


def fused_ssim(img1, img2, padding="same", train=True):
    """submodules/fused-ssim/fused_ssim/__init__.py:34-41 (`FusedSSIMMap.apply(...).mean()`, fused)."""
    C1 = 0.01 ** 2
    C2 = 0.03 ** 2
    assert padding in allowed_padding
    return FusedSSIMMean.apply(C1, C2, img1, img2, padding, train)


KNN_GRID_MIN_POINTS = 2048   # above this the uniform-grid search beats the all-pairs sweep


def distCUDA2(points, method=None):
    """submodules/simple-knn/spatial.cu:15-26: mean squared distance to the 3 nearest neighbours, (P,).
    method: None (automatic), "allpairs" or "grid" -- both are exact and return identical floats."""
    pts = _chk(points, "points")
    if pts.dim() != 2 or pts.shape[1] != 3:
        raise RuntimeError("points must have dimensions (num_points, 3)")
    P = pts.shape[0]
    out = torch.zeros(P, dtype=torch.float32, device=pts.device)
    if P:
        lib = _lib.load()
        stream = torch.cuda.current_stream(pts.device).cuda_stream
        grid = method == "grid" or (method is None and P > KNN_GRID_MIN_POINTS)
        with torch.cuda.device(pts.device):
            if grid:
                nbytes = int(lib.sks_knn3_scratch_bytes(P))
                scratch = torch.empty(nbytes, dtype=torch.uint8, device=pts.device)
                rc = lib.sks_knn3_meandist2_grid(P, pts.data_ptr(), out.data_ptr(), scratch.data_ptr(), nbytes, stream)
                _lib.check(rc, "sks_knn3_meandist2_grid")
            else:
                rc = lib.sks_knn3_meandist2(P, pts.data_ptr(), out.data_ptr(), stream)
                _lib.check(rc, "sks_knn3_meandist2")
    return out
