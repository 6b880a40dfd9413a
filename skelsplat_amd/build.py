"""Builds libskelsplat_hip.so (all hand-written HIP kernels + the C ABI of include/skelsplat_hip.h) for gfx950.

`python -m skelsplat_amd.build` or `skelsplat_amd.build.build()`.  hipcc cross-compiles without a GPU.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libskelsplat_hip.so")
SOURCES = ["sks_raster.hip", "sks_ops.hip", "sks_ssim.hip", "sks_loop.hip"]
# -ffp-contract=off is part of the numeric contract (DESIGN.md "Numerics"): tile lists, n_contrib and the forward
# image must not depend on the compiler's FMA-contraction choices.
FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
         "-fhip-fp32-correctly-rounded-divide-sqrt", "-Wall", "-Wno-unused-function",
         # keep kernel-argument loads in the entry block: sunk into the branch that uses them they become extra
         # dependent scalar round trips in front of the first store of the forward's fill blocks, whose whole life is
         # ~2 us (interleaved A/B of the two builds, tools/ab_libs.sh: forward 53.8 -> 51.9 us, everything else neutral)
         "-mllvm", "-disable-machine-sink"]


def hipcc():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (set HIPCC)")


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "skelsplat_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps if os.path.isfile(d))


def build(force=False, verbose=False):
    if not force and not _stale():
        return LIB
    srcs = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    cmd = [hipcc()] + FLAGS + ["-o", LIB + ".tmp"] + srcs
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
