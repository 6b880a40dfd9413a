"""Generates tests/golden/reference_ssim.npz: the reference's `fused_ssim` Python wrapper, executed.

Run in the build container:  python tests/golden/make_golden_ssim.py

Executed from /root/reference, unmodified: submodules/fused-ssim/fused_ssim/__init__.py -- `FusedSSIMMap` (the "valid" crop
`[:, :, 5:-5, 5:-5]` of the map in forward, the zero-padded `dL_dmap` in backward, `:8-32`) and `fused_ssim()` (C1 = 0.01^2,
C2 = 0.03^2, `map.mean()`, `:34-41`).  Its compiled module `fused_ssim_cuda` (ssim.cu, CUDA) is replaced by a stand-in with
the same two entry points, computed by the reference's OWN PyTorch SSIM -- `utils/loss_utils.py:_ssim` (:277-300), the
function fused-ssim's own test uses as its oracle (tests/test.py:24-54) -- and autograd:
  fusedssim(C1, C2, img1, img2, train)                 -> (map, three per-pixel tensors the wrapper only passes through)
  fusedssim_backward(C1, C2, img1, img2, dL_dmap, ...) -> d sum(map * dL_dmap) / d img1
The fixture therefore pins the WRAPPER's semantics (what row a12's Python half must reproduce: constants, padding modes, the
mean over the cropped map, the gradient reaching img1 only); the kernels' arithmetic is held to the same `_ssim` by the GPU tests.
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    for n in ("tensordict", "cupy", "cupyx", "cupyx.scipy", "cupyx.scipy.ndimage", "plyfile", "cv2"):   # absent here; unused on this path
        sys.modules[n] = types.ModuleType(n)
    sys.modules["tensordict"].TensorDict = dict
    sys.modules["cupyx.scipy.ndimage"].gaussian_filter = None
    sys.modules["plyfile"].PlyData = sys.modules["plyfile"].PlyElement = None
    sys.path.insert(0, REF)
    from utils import loss_utils            # the reference's own conv2d SSIM

    def ssim_map(img1, img2):
        ch = img1.size(-3)
        win = loss_utils.create_window(11, ch).type_as(img1)
        # loss_utils._ssim returns means only: the same expression, map kept (its lines :278-292, called piece by piece)
        F = torch.nn.functional
        mu1, mu2 = F.conv2d(img1, win, padding=5, groups=ch), F.conv2d(img2, win, padding=5, groups=ch)
        s1 = F.conv2d(img1 * img1, win, padding=5, groups=ch) - mu1.pow(2)
        s2 = F.conv2d(img2 * img2, win, padding=5, groups=ch) - mu2.pow(2)
        s12 = F.conv2d(img1 * img2, win, padding=5, groups=ch) - mu1 * mu2
        C1, C2 = 0.01 ** 2, 0.03 ** 2
        return ((2 * mu1 * mu2 + C1) * (2 * s12 + C2)) / ((mu1.pow(2) + mu2.pow(2) + C1) * (s1 + s2 + C2))

    calls = []
    stub = types.ModuleType("fused_ssim_cuda")

    def fusedssim(C1, C2, img1, img2, train):
        calls.append(("fusedssim", float(C1), float(C2), bool(train), tuple(img1.shape)))
        with torch.no_grad():
            m = ssim_map(img1, img2)
        z = torch.zeros_like(m)
        return m, z, z.clone(), z.clone()

    def fusedssim_backward(C1, C2, img1, img2, dL_dmap, dm_dmu1, dm_dsigma1_sq, dm_dsigma12):
        calls.append(("fusedssim_backward", float(C1), float(C2), tuple(dL_dmap.shape), float(dL_dmap.abs().sum())))
        with torch.enable_grad():
            x = img1.detach().clone().requires_grad_(True)
            (ssim_map(x, img2) * dL_dmap).sum().backward()
        return x.grad
    stub.fusedssim, stub.fusedssim_backward = fusedssim, fusedssim_backward
    sys.modules["fused_ssim_cuda"] = stub
    pkg = os.path.join(REF, "submodules", "fused-ssim", "fused_ssim", "__init__.py")
    spec = importlib.util.spec_from_file_location("ref_fused_ssim", pkg)
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    # the map's mean agrees with the reference's own ssim() (size_average): the stand-in IS that function
    g = torch.Generator().manual_seed(0)
    out = {"allowed_padding": np.array(ref.allowed_padding)}
    for tag, shape in (("a", (2, 3, 40, 56)), ("b", (1, 17, 33, 47)), ("c", (1, 1, 12, 16))):
        img1 = torch.rand(shape, generator=g)
        img2 = (img1 + 0.2 * torch.randn(shape, generator=g)).clamp(0, 1)
        assert abs(float(loss_utils.ssim(img1, img2)) - float(ssim_map(img1, img2).mean())) < 1e-6
        out[f"{tag}_img1"], out[f"{tag}_img2"] = img1.numpy(), img2.numpy()
        for padding in ("same", "valid"):
            for train in (True, False):
                x = img1.clone().requires_grad_(True)
                n0 = len(calls)
                val = ref.fused_ssim(x, img2, padding=padding, train=train)
                key = f"{tag}_{padding}_{'train' if train else 'infer'}"
                out[key + "_value"] = np.float64(val.item())
                assert calls[n0][0] == "fusedssim" and calls[n0][1:4] == (0.01 ** 2, 0.03 ** 2, train)
                if train:
                    (3.0 * val).backward()
                    out[key + "_grad"] = x.grad.numpy()
                    # the wrapper hands the backward a FULL-size dL_dmap, zero outside the valid region (:24-27)
                    assert calls[-1][0] == "fusedssim_backward" and calls[-1][3] == shape
        mp = ref.FusedSSIMMap.apply(0.01 ** 2, 0.03 ** 2, img1, img2, "valid", True)
        out[f"{tag}_valid_map_shape"] = np.array(mp.shape)
    out["constants"] = np.array([0.01 ** 2, 0.03 ** 2])
    path = os.path.join(HERE, "reference_ssim.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes;", {k: v.tolist() for k, v in out.items() if k.endswith("_value")})


if __name__ == "__main__":
    main()
