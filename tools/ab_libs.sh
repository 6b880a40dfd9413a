#!/bin/bash
# Interleaved A/B of two builds of the library on one box:  bash tools/ab_libs.sh <libA.so> <libB.so> [dataset] [views]
A=$1; B=$2; DS=${3:-h36m}; V=${4:-4}
for rep in 1 2 3 4; do
  for v in A B; do
    if [ $v = A ]; then export SKS_LIB_OVERRIDE=$PWD/$A; else export SKS_LIB_OVERRIDE=$PWD/$B; fi
    python3 - "$v" "$DS" "$V" <<'PY'
import sys, os
sys.path.insert(0, os.getcwd())
from tools.tune_fwd import setup, run
scene, views, params, dL = setup(sys.argv[2], int(sys.argv[3]))
f, b, tot = run(views, params, dL, 0, iters=100 if int(sys.argv[3]) <= 8 else 20)
print(sys.argv[1], os.path.basename(os.environ["SKS_LIB_OVERRIDE"]), sys.argv[2], f"fwd {f:.1f} us bwd {b:.1f} us step {tot:.1f} us", flush=True)
PY
  done
done
