"""DLT / SVD linear triangulation (reference: triangulation.py:59-67, 111-150), batched over joints.

BASELINE config 1 ("plumbing, no GPU"): the reference solves one 2V x 4 homogeneous system per joint with
np.linalg.svd in a Python loop; here the J systems are one batched torch.linalg.svd (CPU or ROCm tensor)."""
import numpy as np
import torch


def projection_matrices(cameras):
    """P = K [R | t] per camera (triangulation.py:59-67 / 111-119); cameras carry R (camera-to-world), T, K."""
    out = []
    for c in cameras:
        RT = np.hstack((c.R.T, np.asarray(c.T).reshape(3, 1)))
        out.append(np.dot(c.K, RT))
    return np.stack(out, 0)


def triangulate_poses(P_list, poses_2d):
    """triangulation.py:122-150: P_list (V,3,4), poses_2d (V,J,>=2) -> (J,4) homogeneous points with X[3] == 1."""
    P = torch.as_tensor(np.asarray(P_list), dtype=torch.float64)
    x = torch.as_tensor(poses_2d, dtype=torch.float64)[..., :2].to(P.device)
    V, J = x.shape[0], x.shape[1]
    # rows (x * P[2] - P[0]) and (y * P[2] - P[1]) for every view, stacked view-major like the reference
    r0 = x[..., 0:1] * P[:, None, 2, :] - P[:, None, 0, :]          # (V,J,4)
    r1 = x[..., 1:2] * P[:, None, 2, :] - P[:, None, 1, :]
    A = torch.stack([r0, r1], dim=1).reshape(2 * V, J, 4).permute(1, 0, 2)   # (J, 2V, 4)
    _, _, Vt = torch.linalg.svd(A)
    X = Vt[:, -1, :]
    return (X / X[:, 3:4]).cpu().numpy()
