"""Interleaved A/B of the forward fill-block shapes on one box (row-aligned vs linear passes), as quoted in DESIGN.md."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.tune_fwd import setup, run
for dataset, V in (("h36m", 4), ("panoptic", 31)):
    scene, views, params, dL = setup(dataset, V)
    for rep in range(4):
        for name, tune in (("magic modulo", 0), ("integer modulo", 1 << 23)):
            f, b, tot = run(views, params, dL, tune, iters=60 if V == 4 else 15)
            print(f"{dataset} rep{rep} {name}: fwd {f:.1f} us bwd {b:.1f} us step {tot:.1f} us", flush=True)
