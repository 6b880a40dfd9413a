"""Result writer / evaluator compatible with the reference's per-scene .ply files and eval.py.

save_ply: scene/gaussian_model.py:250-281 (fields x,y,z,nx,ny,nz,f_dc_*,f_rest_*,opacity,scale_*,rot_*; binary little
endian, float32) without the plyfile dependency; read_ply_xyz: what eval.py:32-33 reads through open3d;
mpjpe / mpjpe_root_relative: eval.py:123-139."""
import os

import numpy as np


def attribute_names(n_dc, n_rest, n_scale=3, n_rot=4):
    names = ["x", "y", "z", "nx", "ny", "nz"]
    names += [f"f_dc_{i}" for i in range(n_dc)]
    names += [f"f_rest_{i}" for i in range(n_rest)]
    names.append("opacity")
    names += [f"scale_{i}" for i in range(n_scale)]
    names += [f"rot_{i}" for i in range(n_rot)]
    return names


def save_ply(path, gm):
    """gm: object with _xyz (P,3), _features_dc (P,1,C), _features_rest (P,R,C), _opacity (P,1), _scaling, _rotation."""
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    t = lambda x: x.detach().cpu().numpy().astype(np.float32)
    xyz = t(gm._xyz)
    f_dc = t(gm._features_dc.transpose(1, 2).flatten(start_dim=1))
    f_rest = t(gm._features_rest.transpose(1, 2).flatten(start_dim=1))
    cols = np.concatenate((xyz, np.zeros_like(xyz), f_dc, f_rest, t(gm._opacity), t(gm._scaling), t(gm._rotation)), axis=1)
    names = attribute_names(f_dc.shape[1], f_rest.shape[1], gm._scaling.shape[1], gm._rotation.shape[1])
    assert cols.shape[1] == len(names)
    header = ["ply", "format binary_little_endian 1.0", f"element vertex {xyz.shape[0]}"]
    header += [f"property float {n}" for n in names] + ["end_header"]
    with open(path, "wb") as f:
        f.write(("\n".join(header) + "\n").encode("ascii"))
        f.write(np.ascontiguousarray(cols, dtype="<f4").tobytes())


def read_ply_xyz(path):
    """Vertex positions of a binary-little-endian (or ascii) float PLY."""
    with open(path, "rb") as f:
        names, n, fmt = [], 0, None
        while True:
            line = f.readline().decode("ascii").strip()
            if line.startswith("format"):
                fmt = line.split()[1]
            elif line.startswith("element vertex"):
                n = int(line.split()[2])
            elif line.startswith("property"):
                names.append(line.split()[2])
            elif line == "end_header":
                break
        if fmt == "binary_little_endian":
            data = np.frombuffer(f.read(n * len(names) * 4), dtype="<f4").reshape(n, len(names))
        elif fmt == "ascii":
            data = np.loadtxt(f, dtype=np.float32).reshape(n, len(names))
        else:
            raise ValueError(f"unsupported PLY format {fmt}")
    ix = [names.index(k) for k in ("x", "y", "z")]
    return data[:, ix].astype(np.float64)


def mpjpe(pred, gt):
    """eval.py:123-124: mean over frames and joints of the Euclidean error; pred, gt (..., J, 3)."""
    return float(np.mean(np.linalg.norm(np.asarray(gt) - np.asarray(pred), axis=-1)))


def mpjpe_root_relative(pred, gt):
    """eval.py:133-139: both poses translated so that joint 0 is the origin."""
    pred, gt = np.asarray(pred, dtype=np.float64), np.asarray(gt, dtype=np.float64)
    return mpjpe(pred - pred[..., 0:1, :], gt - gt[..., 0:1, :])
