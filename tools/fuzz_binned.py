"""Randomised parity sweep of the binned path against the CPU oracle (GPU box); the cases are tests/fuzz_cases.py's.
    python tools/fuzz_binned.py [cases] [seed0]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from tests.fuzz_cases import run_case

dev = torch.device("cuda:0")
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
bad, t0 = 0, time.time()
for k in range(n_cases):
    try:
        run_case(seed0 + k, dev)
    except AssertionError as e:
        bad += 1
        print("FAIL", str(e)[:500], flush=True)
print(f"{n_cases} cases, {bad} failed, {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
