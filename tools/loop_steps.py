#!/usr/bin/env python3
"""40 chained groups of the sparse training loop (H36M, 4 views): a target for tools/pmc_table.sh / rocprofv3."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from skelsplat_amd.loop import MultiViewLoop
from skelsplat_amd.heatmaps import generate_heatmaps
dev = torch.device("cuda", 0)
wl = bench.WORKLOADS["h36m"]
scene, gm, params = bench.make_scene(torch, wl, dev)
gm.training_setup()
hm = generate_heatmaps(gm._xyz.detach(), gm.get_scaling.detach(), gm._rotation.detach(), torch.tensor(scene.poses_2d, device=dev), scene.cameras)
loop = MultiViewLoop(bench.fresh_model(scene, "h36m", dev), scene.cameras, hm, dataset="h36m", accumulation_steps=4, sparse=True)
loop.run(160)
torch.cuda.synchronize()
