#!/usr/bin/env python3
"""Does the H36M forward's duration depend on WHERE its output planes lie?  Same launch into one big buffer at different byte
offsets (the kernel is timed by the library's own event pairs), several rounds interleaved."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from skelsplat_amd import rasterizer as R, _lib

dev = torch.device("cuda", 0)
scene, gm, params = bench.make_scene(torch, bench.WORKLOADS["h36m"], dev)
views = R.ViewBatch.from_cameras(scene.cameras)
V, P, C, W, H = 4, 17, 17, 1000, 1000
lib = _lib.load()
gbytes, _, _ = _lib.scratch_bytes(V, P, C, W, H, 0)
geom = torch.empty(gbytes, dtype=torch.uint8, device=dev)
radii = torch.empty((V, P), dtype=torch.int32, device=dev)
nc, ni = V * C * H * W, V * H * W
big = torch.empty(nc + ni + (64 << 20) // 4, dtype=torch.float32, device=dev)
base = big.data_ptr()
stream = torch.cuda.current_stream(dev).cuda_stream
offs = [0, 128, 256, 512, 1024, 2048, 4096, 8192, 65536, 1 << 20, (1 << 20) + 4096, 3 << 20]
gaps = [0, 4096, 1 << 20]      # between the colour planes' end and the inverse-depth planes
res = {}
for rnd in range(3):
    for off in offs:
        for gap in gaps:
            pc = base + off
            pi = base + off + 4 * nc + gap
            _lib.prof_enable(True, every=1, kinds=(0,))
            _lib.prof_read(0)
            for _ in range(12):
                rc = lib.sks_forward(V, P, C, W, H, views.viewmatrix.data_ptr(), views.projmatrix.data_ptr(), views.tanfovx, views.tanfovy,
                                     params[0].data_ptr(), params[1].data_ptr(), params[2].data_ptr(), params[3].data_ptr(),
                                     params[4].data_ptr(), None, 1.0, 0, pc, pi, radii.data_ptr(), geom.data_ptr(), None, 0, None, None,
                                     None, stream)
                assert rc == 0
            torch.cuda.synchronize()
            ms, n, q = _lib.prof_read_quantiles(0)
            _lib.prof_enable(False)
            res.setdefault((off, gap), []).append(q[1] * 1e3)
for (off, gap), v in sorted(res.items()):
    print(f"offset {off:8d} B, gap {gap:8d} B: median forward {min(v):6.2f} .. {max(v):6.2f} us over rounds {['%.1f' % x for x in v]}")
