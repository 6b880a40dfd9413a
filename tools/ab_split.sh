#!/bin/bash
# Passes per fill block in the two forms of the step, and the lean fill alone (SKS_FWD_SPLIT=2: no composite launch) (GPU box).
cd "$(dirname "$0")/.."
for tune in 0 0x200 0x300 0x400 0x500; do
  echo "== fused kernel, SKS_FWD_TUNE=$tune"; SKS_FWD_TUNE=$tune python tools/ab_one_call.py 5 2>&1 | grep -v hipGraph
done
for tune in 0x100 0x200 0x300 0x400; do
  echo "== lean fill alone (no composite), SKS_FWD_TUNE=$tune"; SKS_FWD_SPLIT=2 SKS_FWD_TUNE=$tune python tools/ab_one_call.py 5 2>&1 | grep -E "one call:"
done
