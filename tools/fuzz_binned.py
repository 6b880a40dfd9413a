"""Randomised parity sweep of the binned path against the CPU oracle (GPU box); the cases are tests/fuzz_cases.py's.
    python tools/fuzz_binned.py [cases] [seed0] [raster|loss|loop|frames|dropin|ops|onecall]     (loss: the sparse fused-loss step against the dense device path; loop: the production loop against the dense loop)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from tests.fuzz_cases import run_case, run_fused_loss_case, run_loop_case, run_frames_case, run_dropin_case, run_ops_case, run_one_call_case

dev = torch.device("cuda:0")
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
bad, t0, seen = 0, time.time(), []
mode = sys.argv[3] if len(sys.argv) > 3 else "raster"
which = {"raster": run_case, "loss": run_fused_loss_case, "loop": run_loop_case, "frames": run_frames_case, "dropin": run_dropin_case, "ops": run_ops_case, "onecall": run_one_call_case}[mode]
trace = bool(os.environ.get("FUZZ_TRACE"))   # print every seed first and synchronise after it: finds the case behind a GPU fault
for k in range(n_cases):
    try:
        if trace:
            print("seed", seed0 + k, flush=True)
        seen.append(which(seed0 + k, dev))
        if trace:
            torch.cuda.synchronize()
    except AssertionError as e:
        bad += 1
        print("FAIL", str(e)[:500], flush=True)
print(f"{n_cases} cases, {bad} failed, {time.time() - t0:.0f} s")
seen = [x for x in seen if x]
if seen:
    import statistics
    for key in seen[0]:
        vals = sorted(x[key] for x in seen)
        print(f"  {key}: min {vals[0]:.3g}, median {statistics.median(vals):.3g}, max {vals[-1]:.3g}; zero in {sum(v == 0 for v in vals)} cases")
sys.exit(1 if bad else 0)
