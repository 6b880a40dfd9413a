"""Builds libskelsplat_hip.so (all hand-written HIP kernels + the C ABI of include/skelsplat_hip.h) for gfx950.

`python -m skelsplat_amd.build` or `skelsplat_amd.build.build()`.  hipcc cross-compiles without a GPU.
"""
import os
import shutil
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libskelsplat_hip.so")
# source -> extra flags.  sks_ssim.hip: see its header (the SLP vectoriser's packing costs more moves than it saves)
SOURCES = {"sks_raster.hip": [], "sks_ops.hip": [], "sks_ssim.hip": ["-fno-slp-vectorize"], "sks_loop.hip": []}
# -ffp-contract=off is part of the numeric contract (DESIGN.md "Numerics"): tile lists, n_contrib and the forward
# image must not depend on the compiler's FMA-contraction choices.
FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-ffp-contract=off",
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


STAMP = LIB + ".flags"     # the flag set the library was built with (a change of FLAGS / SOURCES is a rebuild)


def _flags_stamp():
    import hashlib
    return hashlib.sha256(repr((FLAGS, sorted(SOURCES.items()))).encode()).hexdigest()


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "skelsplat_hip.h")]
    if any(os.path.getmtime(d) > t for d in deps if os.path.isfile(d)):
        return True
    # a library shipped without its stamp (the snapshot on the GPU box carries both) is taken as it is
    return os.path.exists(STAMP) and open(STAMP).read().strip() != _flags_stamp()


def build(force=False, verbose=False):
    """One builder at a time (every rank of a torchrun job calls this through _lib.load()): an exclusive flock around
    check-compile-link-rename; the link target is a process-unique temporary renamed into place, so a rank never maps a
    half-written file."""
    import fcntl
    with open(LIB + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not _stale():      # another rank built it while this one waited for the lock
                return LIB
            return _build_locked(verbose)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _build_locked(verbose):
    cc = hipcc()
    with tempfile.TemporaryDirectory(prefix="sks_build_") as tmp:
        def compile_one(item):
            src, extra = item
            obj = os.path.join(tmp, src + ".o")
            cmd = [cc] + FLAGS + extra + ["-c", "-o", obj, os.path.join(CSRC, src)]
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            subprocess.check_call(cmd)
            return obj

        with ThreadPoolExecutor(max_workers=4) as pool:   # one translation unit per source, compiled side by side
            objs = list(pool.map(compile_one, SOURCES.items()))
        out = f"{LIB}.{os.getpid()}.tmp"
        cmd = [cc, "--offload-arch=gfx950", "-fPIC", "-shared", "-o", out] + objs
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
    os.replace(out, LIB)
    with open(STAMP, "w") as f:
        f.write(_flags_stamp() + "\n")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
