"""EDT build time on the bunny045 scan at several resolutions (GPU)."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from icp_amd.api import Context
pts = np.fromfile("tests/golden/bunny045.xyz.f32", np.float32).reshape(-1, 3)
ctx = Context(0)
ctx.set_target(pts)
for div in [int(a) for a in sys.argv[1:]] or (64, 128, 300, 512, 768, 1023):
    ctx.edt_build(div)
    ctx.edt_build(div)            # second build runs with the LDS shape adapted to the first
    K = 50 if div <= 512 else 10
    for _ in range(3):
        ctx.edt_rebuild_async()
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(K):
        ctx.edt_rebuild_async()
    ctx.sync()
    dt = (time.perf_counter() - t0) / K
    st = ctx.edt_rebuild_timed(8, 1)
    print(f"div {div:5d}: {dt * 1e3:8.3f} ms/build  {div ** 3 / dt:.3e} voxel/s  kernels (ms) splat {st[0]:.3f} census {st[1]:.3f} zy {st[2]:.3f} x {st[3]:.3f}", flush=True)
