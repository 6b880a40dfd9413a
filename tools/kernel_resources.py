#!/usr/bin/env python3
"""Register / LDS / occupancy table of every kernel of one translation unit (hipcc -Rpass-analysis=kernel-resource-usage):
    python tools/kernel_resources.py [sks_raster.hip] [name regex]"""
import os, re, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
from skelsplat_amd import build as B
src = sys.argv[1] if len(sys.argv) > 1 else "sks_raster.hip"
filt = sys.argv[2] if len(sys.argv) > 2 else "."
flags = [f for f in B.FLAGS if f not in ("-Wall",)] + B.SOURCES.get(src, [])
cmd = [B.hipcc()] + flags + ["-Rpass-analysis=kernel-resource-usage", "-c", "-o", "/dev/null", os.path.join(B.CSRC, src)]
err = subprocess.run(cmd, capture_output=True, text=True).stderr
cur, rows = None, []
for line in err.splitlines():
    m = re.search(r"remark: (?:[^ ]*:\d+:\d+: +)?(.*?) \[-Rpass", line)
    if not m:
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        cur = {"name": t.split(":", 1)[1].strip()}
        rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.split(":", 1)
        cur[k.strip()] = v.strip()
names = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in rows),
                       capture_output=True, text=True).stdout.split("\n")
for r, n in zip(rows, names):
    n = n.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
    if not re.search(filt, n):
        continue
    g = lambda k: r.get(k, "?")
    print("%-58s VGPR %4s AGPR %3s SGPR %4s scratch %4s occ %2s LDS %6s" % (
        n[:58], g("VGPRs"), g("AGPRs"), g("TotalSGPRs"), g("ScratchSize [bytes/lane]"), g("Occupancy [waves/SIMD]"),
        g("LDS Size [bytes/block]")))
