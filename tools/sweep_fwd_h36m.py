import sys, os
sys.path.insert(0, os.getcwd())
from tools.tune_fwd import setup, run
scene, views, params, dL = setup("h36m", 4)
nbytes = 4.0 * scene.H * scene.W * (scene.n_joints + 1) * 4
for rep in range(2):
    for pb in (0, 1, 2, 3, 4, 6):
        f, b, tot = run(views, params, dL, pb << 8, iters=60)
        print(f"passes/block={pb}: fwd {f:6.1f} us ({nbytes/f/1e3:5.0f} GB/s)", flush=True)
    for slots in (2, 4, 8):
        f, b, tot = run(views, params, dL, slots << 26, iters=60)
        print(f"composite slots={slots}: fwd {f:6.1f} us", flush=True)
