"""Result writer / evaluator compatible with the reference's per-scene .ply files and eval.py.

save_ply: scene/gaussian_model.py:250-281 (fields x,y,z,nx,ny,nz,f_dc_*,f_rest_*,opacity,scale_*,rot_*; binary little
endian, float32) without the plyfile dependency; read_ply_xyz: what eval.py:32-33 reads through open3d;
mpjpe / mpjpe_root_relative: eval.py:123-139; evaluate: eval.py:91-171 (the directory walk around them)."""
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


H36M_ACTIVITIES = ("Directions Discussion Eating Greeting Phoning Posing Purchases Sitting SittingDown Smoking Photo "
                   "Waiting Walking WalkDog WalkTogether").split()                     # eval.py:112-114
_S9_BROKEN = ("SittingDown 1", "Waiting 1", "Greeting")                                  # eval.py:27-28, 60-61


def _scene_entries(ply_dir, dataset):
    """eval.py:97-107: scene files of one iteration as sorted [subject, activity, frame-with-extension] triples."""
    entries = os.listdir(ply_dir)
    if dataset == "panoptic":
        parts = [[e.split("_")[0], e.split("_")[1] + "_" + e.split("_")[2], e.split("_")[-1]] for e in entries]
    elif dataset == "occlusion-person":
        parts = [[e.split("_")[0], e.split("_")[1], e.split("_")[-1]] for e in entries]
    else:
        parts = [e.split("_") for e in entries]
    return sorted(parts)


def _gt_poses(gt_path, dataset, absolute, nviews=4):
    """eval.py:54-88: ground-truth poses in sorted subject / activity order (H36M: every 64th frame; S9's three broken
    sequences are left out of the absolute metric)."""
    out = []
    for subject in sorted(os.listdir(gt_path)):
        if not subject.startswith("S"):
            continue
        for activity in sorted(os.listdir(f"{gt_path}/{subject}")):
            if dataset == "h36m":
                if absolute and subject == "S9" and activity in _S9_BROKEN:
                    continue
                out.append(np.load(f"{gt_path}/{subject}/{activity}/poses.npz")["poses"][::64])
            elif dataset == "panoptic":
                out.append(np.load(f"{gt_path}/{subject}/{activity}/poses_filtered_{nviews}.npz", allow_pickle=True)["poses"])
            else:
                out.append(np.load(f"{gt_path}/{subject}/{activity}/poses.npz", allow_pickle=True)["poses3d"])
    return np.concatenate(out, axis=0)


def evaluate(gt_path, output_path, iteration, start_id=0, end_id=10 ** 9):
    """eval.py:91-171 for one iteration: reads <output_path>/point_cloud/iteration_<it>/<scene>.ply (the files the loop's
    results are saved as, scene/__init__.py:112-114) and the dataset's ground truth under gt_path; returns a dict with
    `abs` / `rel` (absolute and root-relative MPJPE over scenes [start_id, end_id)) and, for H36M, the per-activity means
    `abs_activities` / `rel_activities` in the order of H36M_ACTIVITIES (NaN for an activity without scenes).
    Quirk kept: the relative block widens end_id to all predictions (eval.py:133-134, 160-161)."""
    dataset = "panoptic" if "panoptic" in gt_path else "occlusion-person" if "occlusion-person" in gt_path else "h36m"
    ply_dir = f"{output_path}/point_cloud/iteration_{iteration}"
    entries = _scene_entries(ply_dir, dataset)
    res = {}
    for absolute in (True, False):
        gt = _gt_poses(gt_path, dataset, absolute)
        pred, acts = [], []
        for subject, activity, frame in entries:
            if dataset == "h36m" and absolute and subject == "S9" and activity in _S9_BROKEN:
                continue
            pred.append(read_ply_xyz(f"{ply_dir}/{subject}_{activity}_{frame}"))
            acts.append(activity.split(" ")[0])
        pred, acts = np.array(pred), np.array(acts)
        if absolute:
            if end_id > pred.shape[0]:
                end_id = pred.shape[0]
        else:
            if end_id < pred.shape[0]:
                end_id = pred.shape[0]
            gt = gt - gt[:, 0, np.newaxis]
            pred = pred - pred[:, 0, np.newaxis]
        err = np.linalg.norm(gt[start_id:end_id, ...] - pred[start_id:end_id, ...], axis=-1)
        key = "abs" if absolute else "rel"
        res[key] = float(np.mean(err))
        if dataset == "h36m":
            with np.errstate(invalid="ignore"):
                import warnings
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore")
                    res[key + "_activities"] = np.array([np.mean(err[a == acts]) for a in H36M_ACTIVITIES])
    return res
