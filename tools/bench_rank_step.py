#!/usr/bin/env python3
"""One rank's step of the 8-GPU Panoptic run (BASELINE configs[3]) on ONE GPU: bench.extra_rank_step alone, so that
`rocprofv3 --kernel-trace --stats -- python3 tools/bench_rank_step.py` lists exactly its kernels (4 views forward + backward
into the all_gather shard, RCCL all_gather_into_tensor on a 1-rank communicator, sks_mean_views)."""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench

args = argparse.Namespace(steps=int(os.environ.get("STEPS", "200")))
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
print(json.dumps(bench.extra_rank_step(args, torch, dev, torch.cuda.synchronize), indent=1))
