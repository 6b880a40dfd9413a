"""Interleaved A/B of flag-selectable forward variants on ONE box (boxes of the pool differ by +-2 us, which hides
sub-microsecond effects in cross-run comparisons).  Edit the variant list below; include/skelsplat_hip.h lists the tuning
bits (passes / rows per fill block in bits 8..15, SKS_FILL_LINEAR / SKS_FILL_ROWS, composite slots in bits 26..29).
For two BUILDS of the library use tools/ab_libs.sh."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.tune_fwd import setup, run
for dataset, V in (("h36m", 4), ("panoptic", 31)):
    scene, views, params, dL = setup(dataset, V)
    for rep in range(4):
        for name, tune in (("default", 0), ("rows", 1 << 22), ("linear", 1 << 21), ("8 composite slots", 8 << 26)):
            f, b, tot = run(views, params, dL, tune, iters=60 if V == 4 else 15)
            print(f"{dataset} rep{rep} {name}: fwd {f:.1f} us bwd {b:.1f} us step {tot:.1f} us", flush=True)
