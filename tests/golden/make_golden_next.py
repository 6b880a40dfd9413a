"""Generates tests/golden/reference_next.npz by importing and RUNNING the reference's own Python from /root/reference in
the build container (the reference cannot travel to the GPU box; only these small input/output vectors are committed).
Run:  python tests/golden/make_golden_next.py

Pins the "next" rows of SURVEY.md §8f that have a runnable reference here:
  * triangulation.py:111-150  create_projection_matrix(_h36m), triangulate_points_multi_camera, triangulate_poses
  * eval.py:21-171            evaluate(): directory walk, S9 exclusions, absolute and root-relative MPJPE, per-activity means
                              (its prints are np.round(x, 2); the unrounded arguments are recorded through a numpy proxy)
  * utils/general_utils.py:449-498  EarlyStopping / OptEarlyStopping / NotStopping decisions on loss sequences
  * scene/gaussian_model.py:250-281 construct_list_of_attributes + save_ply: the vertex element (field order and bytes)
                              the reference hands to plyfile
Third-party pieces that are absent here and stubbed for the import only: hydra / omegaconf (decorators), open3d
(read_point_cloud serves the arrays this script generated), plyfile (PlyElement.describe captures the structured array).
"""
import io as _io
import contextlib
import os
import sys
import tempfile
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


class _Captured:
    last = None


def install_stubs():
    for n in ("tensordict", "cupy", "cupyx", "cupyx.scipy", "cupyx.scipy.ndimage", "plyfile", "cv2", "open3d", "hydra",
              "hydra.core", "hydra.core.hydra_config", "omegaconf"):
        _stub(n)
    sys.modules["hydra"].__path__ = []
    sys.modules["hydra.core"].__path__ = []
    sys.modules["hydra.core.hydra_config"].HydraConfig = object
    sys.modules["hydra"].main = lambda **k: (lambda f: f)
    sys.modules["omegaconf"].DictConfig = dict
    sys.modules["omegaconf"].OmegaConf = object
    sys.modules["tensordict"].TensorDict = dict
    sys.modules["cupyx.scipy.ndimage"].gaussian_filter = None

    class PlyElement:
        @staticmethod
        def describe(elements, name):
            _Captured.last = (np.array(elements, copy=True), name)
            return ("element", name)

    class PlyData:
        def __init__(self, els):
            self.els = els

        def write(self, path):
            _Captured.path = path

    sys.modules["plyfile"].PlyElement = PlyElement
    sys.modules["plyfile"].PlyData = PlyData
    sys.path.insert(0, REF)


def golden_triangulation(out, rng):
    import triangulation as T
    for tag, V, J, noise in (("h36m4", 4, 17, 3.0), ("pan8", 8, 19, 1.0), ("two", 2, 15, 0.0)):
        Ks, Rs, ts = [], [], []
        pts = rng.normal(0, 500.0, (J, 3)) + np.array([0, 0, 900.0])
        for k in range(V):
            az = 2 * np.pi * k / V + rng.uniform(-0.1, 0.1)
            pos = np.array([4000 * np.cos(az), 4000 * np.sin(az), 1500.0])
            fwd = np.array([0, 0, 900.0]) - pos
            fwd /= np.linalg.norm(fwd)
            right = np.cross(fwd, [0, 0, 1.0]); right /= np.linalg.norm(right)
            down = np.cross(fwd, right)
            R = np.stack([right, down, fwd], 0)
            Ks.append(np.array([[1145.0 + k, 0, 500 + rng.uniform(-15, 15)], [0, 1143.0 - k, 500 + rng.uniform(-15, 15)], [0, 0, 1]]))
            Rs.append(R)
            ts.append((-R @ pos).reshape(3, 1))
        P_list = T.create_projection_matrix_h36m(Ks, Rs, ts)
        names = [f"cam{k:02d}" for k in range(V)]
        P_dict = T.create_projection_matrix({n: K for n, K in zip(names, Ks)}, {n: R for n, R in zip(names, Rs)},
                                            {n: t for n, t in zip(names, ts)})
        assert all(np.array_equal(a, b) for a, b in zip(P_list, P_dict))
        x2d = []
        for P in P_list:
            h = (P @ np.concatenate([pts, np.ones((J, 1))], 1).T).T
            x2d.append(h[:, :2] / h[:, 2:3] + rng.normal(0, noise, (J, 2)) if noise else h[:, :2] / h[:, 2:3])
        x2d = np.stack(x2d)
        conf = rng.uniform(0, 1, (V, J, 1))                       # the reference slices [:2]: a confidence column is ignored
        X = T.triangulate_poses(P_list, torch.tensor(np.concatenate([x2d, conf], -1)))
        X1 = T.triangulate_points_multi_camera(P_list, [x2d[v, 0] for v in range(V)])
        assert np.array_equal(X[0], X1)
        out.update({f"tri_{tag}_K": np.stack(Ks), f"tri_{tag}_R": np.stack(Rs), f"tri_{tag}_t": np.stack(ts)[..., 0],
                    f"tri_{tag}_P": np.stack(P_list), f"tri_{tag}_x2d": x2d, f"tri_{tag}_X": X, f"tri_{tag}_pts": pts})


class _NPProxy:
    """numpy, except that round() records its unrounded argument (evaluate() only prints np.round(x, 2))."""

    def __init__(self, rec):
        self._rec = rec

    def __getattr__(self, k):
        return getattr(np, k)

    def round(self, x, decimals=0):
        self._rec.append(np.array(x, dtype=np.float64))
        return np.round(x, decimals)


def golden_eval(out, rng):
    import importlib
    E = importlib.import_module("eval")
    rec = []
    E.np = _NPProxy(rec)
    preds = {}
    E.o3d.io = types.SimpleNamespace(read_point_cloud=lambda path: types.SimpleNamespace(points=preds[os.path.basename(path)]))
    with tempfile.TemporaryDirectory() as tmp:
        # ---- H36M layout: <gt>/S*/<activity>/poses.npz ['poses'] (frames, 17, 3), every 64th frame is a scene ----
        gt_root = os.path.join(tmp, "data", "h36m")
        outp = os.path.join(tmp, "out_h36m")
        ply_dir = os.path.join(outp, "point_cloud", "iteration_500")
        os.makedirs(ply_dir)
        acts = {"S1": ["Directions", "Walking 1", "Photo"], "S9": ["Greeting", "SittingDown 1", "Eating", "Waiting 1"], "S11": ["Posing"]}
        names, gts, prs = [], [], []
        for subj in sorted(acts):
            for act in sorted(acts[subj]):
                frames = int(rng.integers(130, 200))
                poses = rng.normal(0, 400.0, (frames, 17, 3)) + np.array([0, 0, 900.0])
                os.makedirs(os.path.join(gt_root, subj, act))
                np.savez(os.path.join(gt_root, subj, act, "poses.npz"), poses=poses)
                for f in range(0, frames, 64):
                    name = f"{subj}_{act}_{f:06d}.ply"
                    pred = poses[f] + rng.normal(0, 20.0, (17, 3)) + rng.normal(0, 8.0, (1, 3))
                    preds[name] = pred
                    open(os.path.join(ply_dir, name), "w").close()
                    names.append(name); gts.append(poses[f]); prs.append(pred)
        os.makedirs(os.path.join(gt_root, "not_a_subject"))          # skipped: does not start with 'S'
        with contextlib.redirect_stdout(_io.StringIO()):
            E.evaluate(gt_root, outp, [500], 0, 10 ** 6)
        abs_mean, abs_act, rel_mean, rel_act = rec[:4]
        out.update(eval_h36m_names=np.array(names), eval_h36m_gt=np.stack(gts), eval_h36m_pred=np.stack(prs),
                   eval_h36m_abs=abs_mean, eval_h36m_abs_activities=abs_act, eval_h36m_rel=rel_mean,
                   eval_h36m_rel_activities=rel_act)
        del rec[:]
        # ---- Panoptic layout: poses_filtered_4.npz ['poses'], every frame; scene names S_<a>_<b>_<frame> -------------
        gt_root = os.path.join(tmp, "data", "panoptic")
        outp = os.path.join(tmp, "out_pan")
        ply_dir = os.path.join(outp, "point_cloud", "iteration_500")
        os.makedirs(ply_dir)
        names, gts, prs = [], [], []
        preds.clear()
        for subj, act in (("S1", "pose_1"), ("S1", "pose_2"), ("S3", "haggling_1")):
            frames = 5
            poses = rng.normal(0, 400.0, (frames, 19, 3))
            os.makedirs(os.path.join(gt_root, subj, act))
            np.savez(os.path.join(gt_root, subj, act, "poses_filtered_4.npz"), poses=poses)
            for f in range(frames):
                name = f"{subj}_{act}_{f:04d}.ply"
                pred = poses[f] + rng.normal(0, 15.0, (19, 3)) + rng.normal(0, 5.0, (1, 3))
                preds[name] = pred
                open(os.path.join(ply_dir, name), "w").close()
                names.append(name); gts.append(poses[f]); prs.append(pred)
        with contextlib.redirect_stdout(_io.StringIO()):
            E.evaluate(gt_root, outp, [500], 2, 12)                   # a sub-range: [start_id, end_id)
        out.update(eval_pan_names=np.array(names), eval_pan_gt=np.stack(gts), eval_pan_pred=np.stack(prs),
                   eval_pan_abs=rec[0], eval_pan_rel=rec[1], eval_pan_range=np.array([2, 12]))


def golden_early_stopping(out, rng):
    from utils.general_utils import OptEarlyStopping, EarlyStopping, NotStopping
    f32 = lambda a: [float(np.float32(x)) for x in a]            # the loop hands loss.item() of fp32 tensors
    seqs = {
        "plateau": f32(np.concatenate([np.linspace(1.0, 0.1, 10), np.full(12, 0.1)])),
        "period4": f32(np.concatenate([np.linspace(1.0, 0.2, 9), np.tile([0.2, 0.21, 0.19, 0.205], 5)])),
        "period4_drift": f32(np.concatenate([np.linspace(1.0, 0.2, 9), np.tile([0.2, 0.21, 0.19, 0.205], 5) - 2e-6 * np.arange(20)])),
        "edge": f32([0.5, 0.25, 0.125, 0.0625, 0.5 + 1e-6, 0.25 + 9e-7, 0.125 - 1.1e-6, 0.0625, 0.5, 0.25, 0.125, 0.0625]),
        "noise": f32(rng.uniform(0.1, 1.0, 30)),
        "short": f32([0.3] * 7),
    }
    for name, seq in seqs.items():
        crit = OptEarlyStopping()
        out[f"es_{name}_loss"] = np.array(seq, dtype=np.float64)
        out[f"es_{name}_opt"] = np.array([bool(crit(x)) for x in seq])
        crit3 = OptEarlyStopping(window_size=3, repeat_tolerance=1e-3)
        out[f"es_{name}_opt_w3"] = np.array([bool(crit3(x)) for x in seq])
        pat = EarlyStopping(patience=5, min_delta=1e-3)
        out[f"es_{name}_patience"] = np.array([bool(pat(x)) for x in seq])
        assert not any(NotStopping()(x) for x in seq)


def golden_save_ply(out, rng):
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
        from scene.gaussian_model import GaussianModel
        from utils.graphics_utils import BasicPointCloud
        for key, J in (("h36m", 17), ("panoptic", 19), ("occlusion-person", 15)):
            pts = rng.normal(scale=400.0, size=(J, 3))
            gm = GaussianModel(1)
            gm.create_from_pcd(BasicPointCloud(points=pts, colors=np.zeros((J, 3)), normals=np.zeros((J, 3))),
                               [types.SimpleNamespace(image_name="a")], 5500.0, True, 3.0, J, 1.5, key)
            with torch.no_grad():       # as after an optimisation: every field carries distinct values
                gm._scaling.add_(torch.tensor(rng.normal(0, 0.3, (J, 3)), dtype=torch.float32))
                gm._rotation.add_(torch.tensor(rng.normal(0, 0.2, (J, 4)), dtype=torch.float32))
                gm._opacity.copy_(torch.tensor(rng.normal(2.0, 0.5, (J, 1)), dtype=torch.float32))
            with tempfile.TemporaryDirectory() as tmp:
                gm.save_ply(os.path.join(tmp, "point_cloud", "iteration_500", "S1_Directions_000000.ply"))
            elements, el_name = _Captured.last
            pre = f"ply_{key}_"
            out[pre + "names"] = np.array(elements.dtype.names)
            out[pre + "formats"] = np.array([elements.dtype[n].str for n in elements.dtype.names])
            out[pre + "bytes"] = np.frombuffer(elements.tobytes(), dtype=np.uint8).copy()
            out[pre + "element"] = np.array(el_name)
            out[pre + "attributes"] = np.array(gm.construct_list_of_attributes())
            for f in ("_xyz", "_features_dc", "_features_rest", "_scaling", "_rotation", "_opacity"):
                out[pre + f[1:]] = getattr(gm, f).detach().numpy()
    finally:
        for k, fn in real.items():
            setattr(torch, k, fn)
        torch.Tensor.cuda = real_cuda


def main():
    install_stubs()
    out = {}
    rng = np.random.default_rng(20)
    golden_triangulation(out, rng)
    golden_eval(out, rng)
    golden_early_stopping(out, rng)
    golden_save_ply(out, rng)
    path = os.path.join(HERE, "reference_next.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, {k: np.asarray(v).shape for k, v in out.items()})


if __name__ == "__main__":
    main()
