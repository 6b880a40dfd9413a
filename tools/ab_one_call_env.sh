#!/bin/bash
# sks_forward_backward under its tuning knobs (GPU box): geometry event as a marker packet / on the kernel's dispatch, event
# flags, workgroups per (view, Gaussian) of the backward that runs beside the forward.
cd "$(dirname "$0")/.."
for ext in 0 1; do for wg in 0 2 3 4; do
  echo "== SKS_FB_EXT=$ext SKS_BWD_WG=$wg"; SKS_FB_EXT=$ext SKS_BWD_WG=$wg python tools/ab_one_call.py 5 2>&1 | grep -E "h36m|rank 0" | grep -v hipGraph
done; done
echo "== event flags 0x2, ext"; SKS_FB_EVENT_FLAGS=0x2 python tools/ab_one_call.py 5 2>&1 | grep -E "h36m|rank 0" | grep -v hipGraph
