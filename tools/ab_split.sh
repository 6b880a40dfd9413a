#!/bin/bash
# Passes (rows) per fill block of the forward in the two forms of the API step, interleaved per setting (GPU box): what
# Workspace.tune chooses between.
cd "$(dirname "$0")/.."
for tune in 0 0x200 0x300 0x400 0x500; do
  echo "== SKS_FWD_TUNE=$tune (passes per fill block << 8; 0 = default 2)"; SKS_FWD_TUNE=$tune python tools/ab_one_call.py 5 2>&1 | grep -v hipGraph
done
