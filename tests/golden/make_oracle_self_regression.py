"""Freezes outputs of the CPU oracle (oracle/sks_oracle.c) for a few seeded cases into tests/golden/oracle_self_regression.npz.
These are NOT reference-produced numbers (the reference's CUDA cannot run here); they pin the oracle against silent
regressions and give the GPU tests committed vectors to compare with.  Run: python tests/golden/make_oracle_self_regression.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from tests import util  # noqa: E402

CASES = {"a": dict(seed=21, W=96, H=64, scale_log=4.0), "b": dict(seed=22, W=80, H=80, scale_log=3.5, dataset="panoptic")}


def main():
    out = {}
    for name, kw in CASES.items():
        c = util.make_case(n_views=1, **kw)
        f = util.oracle_forward(c, 0)
        b = util.oracle_backward(c, 0, f, bg=[0.1, 0.2, 0.3])
        for k in ("radii", "xy", "depths", "conic_opacity", "tiles_touched", "point_list", "ranges", "n_contrib", "final_T",
                  "color", "invdepth"):
            out[f"{name}_{k}"] = f[k]
        for k in ("dL_dmeans3D", "dL_dmeans2D", "dL_dopacity", "dL_dscales", "dL_drotations", "dL_dcov3D", "dL_dcolors"):
            out[f"{name}_{k}"] = b[k]
    np.savez_compressed(os.path.join(HERE, "oracle_self_regression.npz"), **out)
    print("wrote oracle_self_regression.npz", len(out), "arrays")


if __name__ == "__main__":
    main()
